/*
 * guardx.h -- C ABI of libguardx_hip.so, the MI355X (gfx950) implementation of
 * the guardX `safe_rl_envs` batched environment step.
 *
 * The reference has NO native boundary for this path (it is pure Python on
 * jax/mjx); the entry points below are what a Python
 * binding of `Engine` (reference safe_rl_envs/safe_rl_envs/envs/engine.py) needs
 * to replace its four jitted callables:
 *
 *   gx_reset       <- Engine.reset        engine.py:454-467 (+ reset_layout/get_layout :433-452)
 *   gx_step        <- Engine.step         engine.py:469-495 (update_data :426-431, mjx_step :659-700)
 *   gx_reset_done  <- Engine.reset_done   engine.py:497-505 (mjx_reset_done :702-731)
 *   gx_rollout     <- the learner's inner loop over step()/reset_done()
 *                     (safe_rl_libX/trpo/trpo.py:466-547) for an open-loop action tape
 *
 * All pointers named `d_*` are DEVICE addresses (hipMalloc / torch tensors on
 * the handle's device), fp32, dense; 16-byte aligned observation buffers take the
 * vector-store path, action rows must be 8-byte aligned.  `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  Nothing here throws;
 * every call returns a gx_status and gx_last_error() describes the last failure
 * on the calling thread.  No call synchronises the device unless documented.
 */
#ifndef GUARDX_H
#define GUARDX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gx_status {
    GX_OK = 0,
    GX_ERR_ARG = 1,         /* bad argument / struct size / alignment */
    GX_ERR_UNSUPPORTED = 2, /* robot or option not implemented */
    GX_ERR_LAYOUT = 3,      /* layout_size <= env_num (engine.py:444 assert) */
    GX_ERR_HIP = 4,         /* a HIP runtime call failed */
    GX_ERR_STATE = 5        /* call order (e.g. step before reset) */
} gx_status;

/* Subset of Engine.DEFAULT (engine.py:98-204) that the hot path reads. */
typedef struct gx_config {
    int32_t struct_size;        /* sizeof(gx_config) */
    int32_t robot;              /* 0 point.xml, 1 swimmer.xml, 2 ant.xml, 3 walker.xml ('robot_base' engine.py:113);
                                 * 4 point.xml with the actuator class defaults (point.xml:7-8) NOT applied to its
                                 * <general> actuators -- the round-1 reading, kept selectable */
    int32_t env_num;            /* envs owned by this handle (local shard) */
    int32_t env_total;          /* env_num of the whole batch (== env_num unsharded) */
    int32_t env_offset;         /* global index of local env 0 */
    uint32_t seed;              /* '_seed'  engine.py:203,216 */
    int32_t num_steps;          /* engine.py:99 */
    int32_t hazards_num;        /* engine.py:195 */
    int32_t lidar_num_bins;     /* engine.py:148 */
    int32_t lidar_alias;        /* engine.py:153 */
    int32_t lidar_max_dist_set; /* engine.py:150, 0 == None */
    float lidar_max_dist;
    float lidar_exp_gain;       /* engine.py:151 */
    float goal_size;            /* engine.py:167 */
    float hazards_size;         /* engine.py:199 */
    float reward_distance;      /* engine.py:174 */
    double goal_keepout;        /* engine.py:166 */
    double hazards_keepout;     /* engine.py:198 */
    double robot_keepout;       /* engine.py:112 */
    double placements_margin;   /* engine.py:104 */
    double extents[4];          /* engine.py:103 */
    int32_t observe_goal_lidar; /* engine.py:119 */
    int32_t observe_goal_comp;  /* engine.py:120 */
    int32_t observe_hazards;    /* engine.py:121 */
    int32_t observe_qpos;       /* engine.py:123 */
    int32_t observe_qvel;       /* engine.py:124 */
    int32_t observe_ctrl;       /* engine.py:128 */
    int32_t observe_vel;        /* engine.py:126 */
    int32_t observe_acc;        /* engine.py:127 */
    int32_t n_candidates;       /* engine.py:263, int(1e6) */
    int32_t physics_steps;      /* engine.py:202 */
    float robot_goal_min_dist;  /* engine.py:571, 3.0 */
    int32_t device;             /* HIP device ordinal ('device_id' engine.py:100) */
    const double* placements;   /* NULL, or (hazards_num+pillars_num+2) x 4 doubles: the placement rectangle
                                 * (xmin,ymin,xmax,ymax) of goal, hazard0.., pillar0.., robot BEFORE the keepout
                                 * shrink -- *_placements / *_locations, engine.py:507-531 */
    /* ---- synthetic extension with NO reference counterpart (BASELINE.json config 5):
     * "pillars", a second class of static circles in the hazard style.  The reference only carries their colour /
     * lidar-group constants (engine.py:38,56) and rejects every config that names them (Bad key).  Here: placed by
     * the layout sampler after the hazards with pillars_keepout; observed by a pseudo-lidar of their own
     * ('pillars_lidar', sorted between 'hazards_lidar' and 'qpos'); cost += sum(pillars_size - min(dist,
     * pillars_size)) after the hazard terms.  pillars_num = 0 is the reference's task. */
    int32_t pillars_num;
    int32_t observe_pillars;
    float pillars_size;
    float robot_rot;            /* 'robot_rot' engine.py:114,342-345: rotation of the robot's root body about z in radians
                                 * (world.py:117); 0 for None (random_rot() returns 0.0, engine.py:330-333).  Was pad_. */
    double pillars_keepout;
} gx_config;

typedef struct gx_engine gx_engine;

const char* gx_last_error(void);
int32_t gx_abi_version(void);
/* identity of the sources and flags this library was built from (guardx_amd/build.py:source_hash) */
const char* gx_build_id(void);
/* `hipcc --version` (HIP + clang version lines) the library was built with; it is part of gx_build_id()'s hash */
const char* gx_build_compiler(void);

gx_status gx_create(const gx_config* cfg, gx_engine** out);
gx_status gx_destroy(gx_engine* e);
int32_t gx_obs_dim(const gx_engine* e);
int32_t gx_act_dim(const gx_engine* e);
/* robot.nq / nv / nu (world.py:435-438) and the action width */
gx_status gx_dims(const gx_engine* e, int32_t* nq, int32_t* nv, int32_t* nu, int32_t* na);

/* Engine.reset: resample the layout pool from the current key, re-initialise
 * every env, write d_obs (env_num x obs_dim).  Asynchronous on `stream`. */
gx_status gx_reset(gx_engine* e, float* d_obs, void* stream);

/* Size of the valid-layout pool of the last gx_reset.  SYNCHRONISES `stream`
 * of that reset.  Returns GX_ERR_LAYOUT if *out <= env_total (engine.py:444). */
gx_status gx_layout_size(gx_engine* e, int32_t* out);
/* Deferred form of the same assert: smallest pool size over all gx_reset calls since the previous
 * gx_layout_size_min (synchronises the last reset only).  Lets a driver queue epochs without a host
 * round trip per reset and still honour engine.py:444. */
gx_status gx_layout_size_min(gx_engine* e, int32_t* out);

/* Engine.step.  d_action (env_num x act_dim) -> d_obs (env_num x obs_dim),
 * d_reward, d_cost, d_done (env_num each; done is 0.f/1.f), d_qacc
 * (env_num x nv, may be NULL).  No auto-reset. */
gx_status gx_step(gx_engine* e, const float* d_action, float* d_obs, float* d_reward,
                  float* d_cost, float* d_done, float* d_qacc, void* stream);

/* Engine.step plus, in the same launch, what a reset_done() right after it returns: d_obs_rd (env_num x obs_dim)
 * = d_obs with the rows of the envs this step finished replaced by their re-initialised observation
 * (engine.py:497-505, same key).  The state is NOT re-initialised by this call: gx_reset_done_commit() requests
 * that, and the next launch on this handle installs it (no launch of its own, so the learner's
 * `step(); if done.any(): reset_done()` pair costs one kernel).  *speculated = 0 when the handle runs the
 * thread-per-env kernels (env_num > 16384): d_obs_rd is then left unwritten and the caller uses gx_reset_done. */
gx_status gx_step_rd(gx_engine* e, const float* d_action, float* d_obs, float* d_reward, float* d_cost,
                     float* d_done, float* d_qacc, float* d_obs_rd, int32_t* speculated, void* stream);
/* gx_step / gx_step_rd with the outputs addressed inside ONE caller-owned allocation ("slab") of consecutive output sets:
 * set `slot` starts at d_slab + slot * gx_step_set_floats() floats (d_slab 16-byte aligned) and holds, every piece 16-byte
 * aligned (Dp = obs_dim rounded up to 4, Np = env_num rounded up to 4):
 *   obs [N][D] at 0 | obs_rd [N][D] at N*Dp | reward [N] at 2*N*Dp | cost [N] at +Np | done [N] at +2*Np | qacc [N][nv] at +3*Np
 * flags: bit 0 = write qacc, bit 1 = also evaluate obs_rd (= gx_step_rd; *speculated as there).  The host hands out
 * views of a set and never reuses it (engine.py:495 returns fresh buffers); one pointer and an index per call instead
 * of six addresses is what the learner-driven step()+reset_done() loop needs at env_num = 2000, where it is host bound. */
gx_status gx_step_set_floats(const gx_engine* e, int64_t* floats);
gx_status gx_step_slab(gx_engine* e, const float* d_action, float* d_slab, int32_t slot, int32_t flags,
                       int32_t* speculated, void* stream);
/* Engine.reset_done for the step just made through gx_step_rd (speculated == 1): host-only, idempotent.
 * GX_ERR_STATE if the last hot-path call was anything else. */
gx_status gx_reset_done_commit(gx_engine* e);

/* Engine.reset_done: rows of envs whose last done was > 0 are re-initialised
 * and their obs rows replaced; other rows are copied from d_obs_in (which may
 * alias d_obs_out). */
gx_status gx_reset_done(gx_engine* e, const float* d_obs_in, float* d_obs_out, void* stream);

/* T fused step()+reset_done() iterations driven by an action tape
 * d_actions[T][env_num][act_dim].  Outputs are time-major: d_obs[T][env_num][obs_dim]
 * is the observation the learner sees AFTER reset_done (trpo.py:547), d_reward /
 * d_cost / d_done [T][env_num]. */
gx_status gx_rollout(gx_engine* e, int32_t T, const float* d_actions, float* d_obs,
                     float* d_reward, float* d_cost, float* d_done, void* stream);

/* gx_rollout writing the learner hand-off layout directly: d_packed[T][env_num][W], W = gx_packed_width() =
 * obs_dim + act_dim + 3, row = (obs | action | reward, cost, done) -- what TRPOBufferX stores per step
 * (safe_rl_libX/trpo/trpo.py:34-42,49-64), ready for ONE all-gather per epoch with no pack pass. */
gx_status gx_rollout_packed(gx_engine* e, int32_t T, const float* d_actions, float* d_packed, void* stream);
int32_t gx_packed_width(const gx_engine* e);

/* ---- tape hand-off (multi-GPU; every robot) --------------------------------------------------------------
 * The packed row is 4 * (obs_dim + act_dim + 3) bytes per env-step (192 B for Goal_Point_8Hazards); an all-gather of it
 * over xGMI takes longer than the epoch that produced it.  The observation is a function of 80 B of state, so the
 * stepping rank runs only the serial dynamics pass of gx_rollout and hands out that tape; every rank that needs the
 * rollout (safe_rl_libX/trpo/trpo.py:34-42: obs, act, rew, cost, done) runs the observation pass on the gathered
 * tapes and gets the packed rows of gx_rollout_packed, bit for bit.  All ranks sample identical layout pools (shared
 * key, engine.py:263), so the pool rows a tape's reset_done events refer to are local on every rank.
 *   d_shard: gx_tape_floats() floats = [ tape | layouts at entry | entry records ], 16-byte aligned; all-gather it as
 *            is.  A tape row is qpos | qvel | action | ONE word for done, the layout row in effect and the layout row
 *            reset_done installed (c >= 0: not done, layout row c - 1 of the pool in effect -- 0: the layout at entry;
 *            -1: done and nothing installed; c <= -2: done, row -(c + 2) installed; the layout in effect during a step
 *            that finished the env is what the rows before it say):
 *            9 floats (36 B) per env-step for the Point, 13 for the Swimmer, 32 for the Ant and 38 for the Walker
 *            (including, for those two, the row of the pool's fake-step table a reset_done observation is read from);
 *            rows are 4-byte aligned only; the
 *            tape is rounded up to a multiple of 4 floats so that the layouts behind it stay 16-byte aligned; the
 *            observation pass re-derives the pose, ctrl and the reward from consecutive rows.  One physics step per
 *            control step and no observe_vel / observe_acc only (GX_ERR_UNSUPPORTED otherwise).
 *   token:   names the layout pool in effect; gx_expand_tape (on any engine of the same configuration and key
 *            history, e.g. the other ranks') must be CALLED before the second gx_reset after the rollout --
 *            the engines keep three pools for that; GX_ERR_STATE afterwards.  The next sampler that reuses the
 *            pool is ordered behind the expansion by an event, whatever stream it ran on. */
gx_status gx_tape_floats(const gx_engine* e, int32_t T, int64_t* tape, int64_t* layouts, int64_t* entry);
gx_status gx_rollout_tape(gx_engine* e, int32_t T, const float* d_actions, float* d_shard, int64_t* token,
                          void* stream);
gx_status gx_expand_tape(gx_engine* e, int32_t T, const float* d_shard, int64_t token, float* d_packed,
                         void* stream);
/* The same over the shards of n_shards ranks in ONE launch: shard s at d_shards + s * stride_floats (the all-gathered
 * buffer as it is; a multiple of 4 floats), its packed rows at d_packed + s * packed_stride_floats.  Eight separate
 * 400 000-row launches each pay their own ramp and tail; one 3.2 M-row launch runs at the rate of the large-batch
 * step kernel. */
gx_status gx_expand_tapes(gx_engine* e, int32_t T, const float* d_shards, int64_t stride_floats, int32_t n_shards,
                          int64_t token, float* d_packed, int64_t packed_stride_floats, void* stream);

/* ---- sharded layout sampling (multi-GPU, OPTIONAL: a second collective on the reset path) -----------------
 * reset()'s rejection sampler (engine.py:433-444, 546-621) draws 1e6 independent candidates (candidate c from
 * split(key, 1e6)[c]) and keeps the valid ones in candidate order.  With envs sharded over W ranks every rank would
 * sample all of them.  Instead: rank r calls gx_sample_shard(r, W) for candidates [r 1e6 / W, (r+1) 1e6 / W) of the
 * reset about to happen, all-gathers the exported rows and counts, and gx_reset_from_shards installs the concatenation
 * (shard after shard = candidate order) as the pool and re-initialises the envs: layout_size, the pool rows and every
 * later randint draw equal the unsharded gx_reset's.  The layout prefetch must be off (gx_set_prefetch(e, -1)).
 *   d_rows:  [cap][goal, hazards.., pillars.., robot][2] floats, this shard's valid layouts in candidate order
 *   d_count: their number (device int32; may exceed cap, in which case gx_reset_from_shards leaves a negative
 *            layout_size and the layout check fails)
 *   d_rows_all / d_counts: the all-gathered [W][cap][..][2] / [W] */
gx_status gx_sample_shard(gx_engine* e, int32_t shard, int32_t n_shards, float* d_rows, int32_t cap, int32_t* d_count,
                          void* stream);
gx_status gx_reset_from_shards(gx_engine* e, const float* d_rows_all, const int32_t* d_counts, int32_t n_shards,
                               int32_t cap, float* d_obs, void* stream);

/* ---- the same riding on the rollout hand-off: ONE collective per epoch (the default multi-GPU path since round 4) --
 * The key of a later reset is known now: this key advanced by one split per step() (engine.py:431), the number of steps
 * between resets being the learned horizon of gx_set_prefetch.  The pipeline guardx_amd/dist.py:TapeHandoff runs
 * (epochs and resets counted from 0, epoch k = gx_reset(k) + its rollout):
 *   epoch k, right after gx_reset(k) and BEFORE its rollout is launched (since round 5; rounds 4-5: after the epoch's
 *   collective had been issued -- same keys, same blocks, but the sampler could then only start once the dynamics pass
 *   had ended):
 *       gx_sample_shard_ahead(resets_ahead = 3) -- rank r samples ITS candidates of reset(k + 3) on the engine's side
 *       stream, beside epoch k's dynamics pass.  (resets_ahead counts from the LAST reset: the key is advanced by
 *       resets_ahead * horizon - steps_since_reset: three horizons when called before the epoch's steps, two when called
 *       after them -- the same reset either way.)  The export block -- [count, key0, key1, shard | n_shards << 16 | rows cap x
 *       (goal, hazards.., pillars.., robot) x 2 floats] -- is the tail of the buffer that will carry epoch k + 1's tape.
 *   epoch k + 1: gx_shard_join, then the all-gather of [tape k + 1 | block] delivers every rank's block.
 *   epoch k + 2: gx_install_shards(ticket of that call), on the stream that waited for the collective, AFTER
 *       gx_reset(k + 2) and BEFORE gx_reset(k + 3): it writes the pool slot the NEXT gx_reset takes ((cur + 1) % 3), so
 *       exactly one install may lie between two consecutive resets -- an install made one reset early overwrites a
 *       pool that has not been taken yet, one made late lands in the wrong slot; either way the key check at the reset
 *       fails and that reset falls back to sampling inline (results unchanged, the sharing lost).
 *   gx_reset(k + 3) takes the installed pool exactly like a prefetch hit.
 * A reset whose key has no installed pool (the first three of a hand-off, a changed episode length) samples all
 * candidates inline like a prefetch miss: layouts, observations and every later draw are the reference's either way
 * (engine.py:433-452).  The sampler's work per GPU is 1/W of the reference's and no second collective exists.
 *   gx_set_layout_source(e, 1): gx_reset launches no prefetch sampler of its own; 0 (default) restores it.
 *   gx_shard_block_floats:  floats of one export block of capacity `cap` rows (a multiple of 4).
 *   gx_sample_shard_ahead:  resets_ahead >= 1 (3 in the pipeline above).  d_block 16-byte aligned.  GX_ERR_STATE when the
 *                           engine has no horizon (gx_set_prefetch(e, -1)) or more steps have been made since the last
 *                           reset than resets_ahead horizons cover: a state all ranks share -- skip the block on every
 *                           rank (TapeHandoff does), the reset it was meant for samples inline.  The sampler starts
 *                           behind everything already queued on `stream`; gx_shard_join makes `stream` wait for the block
 *                           (call it before the collective that sends it, and before freeing the buffer).
 *                           *ticket names the call: every rank makes the same calls in the same order, so the
 *                           ticket of the blocks that arrive with a collective is the local one of that epoch.
 *   gx_install_shards:      d_blocks = block of shard 0, shard s at d_blocks + s * stride_floats; `ticket` = the
 *                           gx_sample_shard_ahead call whose key / n_shards / cap these blocks belong to (the engine
 *                           remembers its last four; GX_ERR_STATE otherwise).  A block of another key, shard or world
 *                           size, or count > cap, leaves layout_size < 0: the reset that takes the pool fails its
 *                           layout check (engine.py:444) instead of using it. */
/* A stream of the engine's device -- one per device and process, created on first use, never destroyed (valid for the
 * life of the process) -- for throughput work the caller runs beside the stepping: the hand-off's gx_install_shards /
 * gx_expand_tapes.  Ordinary priority (round 5: with the hand-off's streams in flight a queue of another priority class
 * delayed everything behind the dynamics pass by ~100 us per epoch); tested at creation not to share
 * the default stream's hardware queue (GX_STREAM_CHECK=0 skips the test).  The call also tells the engine that it now
 * works beside a hand-off: from here on its layout sampler runs on a stream of the same priority class (created now, before
 * the caller makes its collective's streams) instead of its least-priority one. */
gx_status gx_aux_stream(gx_engine* e, void** stream);
/* The same, but the device's stream is REPLACED by a new one first (the old one stays alive for the life of the process, so
 * the new one lands on another hardware queue): for a caller that measured the stream sharing a queue with the stream of
 * its collective (guardx_amd.dist.TapeHandoff probes this once, at construction). */
gx_status gx_aux_stream_renew(gx_engine* e, void** stream);
gx_status gx_set_layout_source(gx_engine* e, int32_t source);
gx_status gx_shard_block_floats(const gx_engine* e, int32_t cap, int64_t* floats);
gx_status gx_sample_shard_ahead(gx_engine* e, int32_t shard, int32_t n_shards, int32_t resets_ahead, float* d_block,
                                int32_t cap, int64_t* ticket, void* stream);
gx_status gx_shard_join(gx_engine* e, void* stream);
gx_status gx_install_shards(gx_engine* e, int64_t ticket, const float* d_blocks, int64_t stride_floats, int32_t n_shards,
                            int32_t cap, void* stream);

/* ---- closed-loop fused rollout with an on-device policy ------------------
 * `ac.step(o)` of MLPActorCritic(hidden_sizes=(64,64), tanh) (safe_rl_libX/trpo/trpo_core.py:110-173)
 * evaluated inside the persistent rollout kernel: per step  a ~ N(mu_net(o), exp(log_std)), logp,
 * v_net(o); env.step(a); reset_done().  d_params (device, fp32, torch layouts):
 *   mu_net {W1[64][D] b1[64] W2[64][64] b2[64] W3[A][64] b3[A]}  v_net {.. W3[1][64] b3[1]}  log_std[A]
 * The action noise is a counter-based stream keyed by `seed` (Threefry block (env, step) -> Box-Muller),
 * not torch's generator.  Outputs are time-major: d_obs_in[T][N][D] is the observation the policy saw at
 * step t (what the learner stores), d_act/d_mu [T][N][A], d_logp/d_val/d_reward/d_cost/d_done [T][N];
 * d_obs_last[N][D], d_val_last[N] = o_T and V(o_T) for the bootstrap; d_logstd[A] = log(std). */
typedef struct gx_policy {
    int32_t struct_size;   /* sizeof(gx_policy) */
    int32_t hidden;        /* 64 (the reference default, trpo.py:606 --hid), 128, 192, 256.  One fused launch per call for
                            * the light robots (Point, Swimmer) at their default observation width: 64 with the networks in
                            * LDS, 128 with the hidden layers resident in registers as MFMA operands, 192 / 256 with them
                            * streamed from an L2-resident transposed copy.  Otherwise (Ant, Walker, other observation widths,
                            * gx_set_policy_impl 1 | 2 | 3) two launches per control step: policy over all envs, then the
                            * fused step + reset_done.  Same arithmetic, same results as the checker, whichever runs. */
    const float* d_params; /* device pointer, layout above */
    uint32_t seed[2];
} gx_policy;
gx_status gx_rollout_policy(gx_engine* e, int32_t T, const gx_policy* pol, const float* d_obs0,
                            float* d_obs_in, float* d_act, float* d_logp, float* d_val, float* d_mu,
                            float* d_reward, float* d_cost, float* d_done, float* d_obs_last,
                            float* d_val_last, float* d_logstd, void* stream);
/* How the two hidden layers are evaluated: 0 auto (= 2), 1 VALU fmaf chains with one wave per
 * workgroup, 2 v_mfma_f32_16x16x4_f32 tiles with 16 envs per workgroup, 3 the step-wise form the wider networks use
 * (two launches per control step; available at hidden = 64 as a cross-check).  Bit-identical results. */
gx_status gx_set_policy_impl(gx_engine* e, int32_t impl);
gx_status gx_math_probe2(int32_t n, const float* d_x, float* d_log, float* d_tanh, void* stream);

/* Test / checkpoint support: env-major HOST arrays (any may be NULL).
 *  qpos[N*nq] qvel[N*nv] pose0[N*4] pose1[N*2] objs[N*(1+H)*2] done0[N] done1[N]
 *  steps[N] key[2] hist[1].  Synchronous. */
gx_status gx_get_state(gx_engine* e, float* qpos, float* qvel, float* pose0, float* pose1,
                       float* objs, float* done0, float* done1, float* steps,
                       uint32_t* key, int32_t* hist);
gx_status gx_set_state(gx_engine* e, const float* qpos, const float* qvel, const float* pose0,
                       const float* pose1, const float* objs, const float* done0,
                       const float* done1, const float* steps, const uint32_t* key,
                       const int32_t* hist);
/* rows of the valid-layout pool ((H+2)*2 floats each: goal, hazards.., robot) */
gx_status gx_get_pool(gx_engine* e, float* pool, int32_t max_rows, int32_t* got);

/* Layout-pool prefetch: after each gx_reset the pool for the NEXT reset (the key advanced
 * by `steps` step() calls, engine.py:431) is sampled on a low-priority side stream while the
 * epoch runs; a reset whose key matches uses it, otherwise it samples inline.  Results are
 * identical either way.  steps >= 0: fixed prediction; -1: no prefetch; -2 (default): predict the
 * number of steps made between the last two resets (cfg.num_steps before the second reset) -- the
 * learners reset every max_ep_len steps (trpo.py:453,517), which need not equal num_steps. */
gx_status gx_set_prefetch(gx_engine* e, int32_t steps);
/* prefetched pools used / discarded so far, and the current prediction */
gx_status gx_prefetch_stats(const gx_engine* e, int32_t* hits, int32_t* misses, int32_t* horizon);

/* Kernel family used by step / rollout: 0 = auto (up to 16384 envs -- Ant 27000, Walker 16000, the measured
 * crossovers -- lane-group kernels, and for rollouts of 8+ steps the two-kernel form -- serial dynamics tape, then one
 * thread per (step, env) observation row; thread-per-env kernels above those sizes), 1 = force thread-per-env,
 * 2 = force lane-group,
 * 3 = two-kernel rollouts at any T where supported (lane-group otherwise).
 * Results are bit-identical either way (tests/test_gpu_parity.py). */
gx_status gx_set_path(gx_engine* e, int32_t mode);

/* ---- learner-side rollout buffer on device -------------------------------
 * Counterparts of TRPOBufferX (safe_rl_libX/trpo/trpo.py:24-146); buffers are env-major
 * (env_num, max_ep_len, .) fp32 device arrays owned by the caller.
 *   gx_buffer_store     <- TRPOBufferX.store        trpo.py:49-64   (7 strided writes -> 1 launch)
 *   gx_gae_finish_path  <- TRPOBufferX.finish_path  trpo.py:66-119  (GAE-lambda + rewards-to-go for
 *                          the envs whose d_done == 1, all envs when d_done is NULL; d_path_start is
 *                          advanced to ptr for them; no host sync, no per-env Python loop)
 *   gx_adv_normalize    <- the per-env advantage normalisation in TRPOBufferX.get  trpo.py:131-135
 * CPOBufferX (safe_rl_libX/cpo/cpo.py:22-175) is the same with a second (cost, cost_val) channel:
 * call gx_gae_finish_path twice, advancing d_path_start only on the second call. */
gx_status gx_buffer_store(int32_t env_num, int32_t max_ep_len, int32_t ptr, int32_t obs_dim,
                          int32_t act_dim, const float* d_obs, const float* d_act, const float* d_rew,
                          const float* d_val, const float* d_logp, const float* d_mu,
                          const float* d_logstd, float* d_obs_buf, float* d_act_buf, float* d_rew_buf,
                          float* d_val_buf, float* d_logp_buf, float* d_mu_buf, float* d_logstd_buf,
                          void* stream);
gx_status gx_gae_finish_path(int32_t env_num, int32_t max_ep_len, int32_t ptr, const float* d_rew_buf,
                             const float* d_val_buf, const float* d_last_val, const float* d_done,
                             int32_t* d_path_start, double gamma, double lam, float* d_adv_buf,
                             float* d_ret_buf, int32_t advance_path_start, void* stream);
/* GAE-lambda over the time-major outputs of gx_rollout / gx_rollout_policy ([T][env_num] arrays): what
 * store() + finish_path() at every done step + the closing finish_path() of trpo.py:466-547 produce, in
 * one launch.  A path ends at every step with d_done == 1 (bootstrap 0) and at the end of the tape
 * (bootstrap d_last_val[env]; pass zeros to mirror trpo.py:506-515). */
gx_status gx_gae_rollout(int32_t env_num, int32_t T, const float* d_rew, const float* d_val,
                         const float* d_done, const float* d_last_val, double gamma, double lam,
                         float* d_adv, float* d_ret, void* stream);
/* scale != 0: (x - mean) / std (reward advantage); scale == 0: x - mean (CPO cost advantage,
 * safe_rl_libX/cpo/cpo.py:158-162) */
gx_status gx_adv_normalize(int32_t env_num, int32_t max_ep_len, float* d_adv_buf, int32_t scale,
                           void* stream);

/* Profiling aid: while d_stamps (device, ceil(env_num/4) x 8 uint64) is set, the lane-group kernels record the
 * shader clock (s_memtime) of each workgroup at: 0 entry, 1 state+action loads issued, 2 loads arrived; for step
 * t* = min(T-1, 100): 3 start, 4 dynamics done, 5 lidar/cost exchange done, 6 end of step; 7 state stored.
 * NULL switches it off (the default). */
gx_status gx_debug_stamps(gx_engine* e, uint64_t* d_stamps);

/* Device-math probe (tests): s,c = sincos(x); at2 = atan2(y,x); ex = exp(x). */
gx_status gx_math_probe(int32_t n, const float* d_x, const float* d_y, float* d_s,
                        float* d_c, float* d_at2, float* d_ex, void* stream);
/* Device PRNG probe (tests): out[2n] = jax.random.split(key, n) computed on device. */
gx_status gx_split_probe(const uint32_t* key, int32_t n, uint32_t* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
