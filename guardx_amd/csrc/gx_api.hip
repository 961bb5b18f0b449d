// gx_api.hip -- C ABI (include/guardx.h) over the HIP kernels.  Host side only.
//
// The handle owns the SoA environment state and the layout pool; observation,
// reward, cost, done and action buffers belong to the caller (torch tensors).
// Reference behaviour cited as engine.py:NNN
// (/root/reference/safe_rl_envs/safe_rl_envs/envs/engine.py).
#include "../../include/guardx.h"
#include "gx_kernels.h"
#include "gx_robot.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

using namespace gx;

static thread_local std::string g_err;

static gx_status fail(gx_status st, const std::string& msg)
{
    g_err = msg;
    return st;
}

gx_status gx_fail_msg(gx_status st, const char* msg) { return fail(st, msg); } // for gx_gae.hip

#define GX_HIP(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(GX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));    \
    } while (0)

struct gx_engine {
    gx_config cfg;
    Params p;
    SampleParams sp;
    float phase1_pass_estimate = 0.f; // of the candidates that reach phase 1 (gx_create: picks the sampler's form)
    DevBuffers b;
    int nobj_total;
    uint32_t key[2];
    int hist;            // number of step() calls so far, saturating at 2
    bool have_reset;
    int* h_layout_size;  // pinned
    hipEvent_t layout_ev;
    bool layout_pending;
    int device;
    int nq, nv, nu, na, ndyn; // robot.nq/nv/nu (world.py:435-438), action width, float4s of state
    float4* haz_bounds;  // device copy of the per-hazard placement bounds (or null)
    uint32_t policy_steps; // ac.step() calls made through gx_rollout_policy (noise counter)
    int policy_impl;       // 0 auto (MFMA), 1 VALU fmaf chains, 2 fp32 MFMA tiles
    int path_mode;       // 0 auto, 1 thread-per-env kernels, 2 lane-group kernels
    // double-buffered layout pools + side stream: the pool of the NEXT reset() is sampled
    // while the current epoch is being stepped (the key chain is data-independent)
    // a ring of THREE: the pool of epoch k is overwritten by the prefetch launched at reset(k+2), so a dynamics tape
    // handed to other ranks (gx_rollout_tape) can be expanded there (gx_expand_tape, which reads the pool rows its
    // reset_done events refer to) during the whole of epoch k+1, overlapped with its all-gather
    static const int kPools = 3;
    Pool pools[kPools];
    int cur;                 // pool the envs are drawn from
    hipStream_t side[kPools]; // prefetch samplers: pool i is sampled on side[i % n_side] (least priority)
    // the sampler's stream while the engine works beside a tape hand-off (its own aux stream in use, or layouts coming
    // from the ranks' shards): ORDINARY priority -- see side_stream_priority()
    hipStream_t side_ord = nullptr;
    hipEvent_t side_switch = nullptr;
    bool aux_in_use = false;
    int sampler_class = 0;    // 0: the last sampler went onto side[], 1: onto side_ord
    int n_side;
    hipEvent_t pool_ready[kPools]; // recorded on the sampling stream when pool i is complete
    hipEvent_t pool_free[kPools];  // recorded on the caller's stream when pool i is no longer read
    hipEvent_t expand_ev[kPools];  // recorded behind the last gx_expand_tape that read pool i
    bool expand_pending[kPools];
    uint32_t pool_gen[kPools];     // bumped whenever a sampler is launched into pool i (tape tokens)
    bool pf_valid;            // pools[(cur+1)%kPools] holds (or will hold) the pool for key pf_key
    uint32_t pf_key[2];
    int prefetch_steps;       // predicted step() calls between resets; -1 disables prefetch; -2 = learn it:
                              // the number of steps between the last two resets (cfg.num_steps before that) --
                              // the learners reset every max_ep_len steps, whatever num_steps says
    int steps_since_reset;    // key advances since the last gx_reset
    int last_interval;        // ... between the last two gx_reset calls (0: unknown)
    int pf_hits, pf_misses;
    bool last_policy = false; // the last hot-path call was gx_rollout_policy
    // speculated reset_done (gx_step_rd): b.rd_j holds the layout rows reset_done would install for the envs the
    // last step finished; gx_reset_done_commit() only sets pending_commit, the next launch installs them
    unsigned long long* stamps = nullptr; // gx_debug_stamps
    // two-kernel rollout (gx_split_rollout.inl): dynamics tape [T][N][W] and the layout snapshot of its launch
    float* tape = nullptr;
    size_t tape_cap = 0;      // floats
    float4* obj0 = nullptr;
    hipEvent_t pf_phase1 = nullptr;  // prefetch sampler: phases 0 and 1 done (the observation pass is held until then)
    bool pf_phase1_pending = false;
    bool spec_valid = false;
    bool pending_commit = false;
    // per-step layout keys of the fused rollouts: a ring of staging slots in ONE pinned, device-visible allocation
    // (stage_rollout_keys)
    uint4* h_keys = nullptr;
    int keys_cap = 0;               // keys (steps) per slot
    int keys_slots = 0;
    int keys_next = 0;
    std::vector<hipStream_t> keys_streams; // streams that were handed a slot in the current lap of the ring
    // step-wise policy rollout (hidden widths beyond the fused kernel's): transposed weights, current observation
    float* pol_wt = nullptr;
    size_t pol_wt_cap = 0;
    float* pol_cur = nullptr;
    // ---- sharded layout sampling ----
    // gx_sample_shard -> gx_reset_from_shards: the pool and key the last shard was sampled for (claim_pool done there)
    bool rs_sampled = false;
    int rs_pool = -1;
    uint32_t rs_key[2] = {0, 0};
    // piggy-backed form (gx_sample_shard_ahead / gx_install_shards): layout_source 1 = the pool of the next reset() is
    // installed from the ranks' export blocks, gx_reset launches no prefetch sampler of its own
    int layout_source = 0;
    Pool shard_scratch;            // the shard sampler's own working pool (the ring's three are all in use)
    bool shard_scratch_ok = false;
    int shard_scratch_cap = 0;     // candidates it holds
    hipEvent_t shard_dep = nullptr, shard_done = nullptr;
    bool shard_inflight = false;
    struct ShardJob { int64_t ticket; uint32_t k0, k1; int n_shards, cap; };
    static const int kJobs = 4;
    ShardJob jobs[kJobs];          // the last four gx_sample_shard_ahead calls, slot = ticket % kJobs
    int64_t next_ticket = 1;
};

// the pending reset_done (if any) is consumed by the launch about to be made / dropped by reset()
static int take_commit(gx_engine* e)
{
    const int c = e->pending_commit ? 1 : 0;
    e->pending_commit = false;
    e->spec_valid = false;
    return c;
}

// env_num at which the lane-group form of a step beats the thread-per-env form.  Measured crossovers of the two
// families (fused step incl. reset_done, tools/history/debug/legs_large.py, round 3 -- after the legs' lanes stopped
// replicating work): Point / Swimmer 16384 envs; Ant ~27 k (lane-group 38.6 us at 24576 against ~44 us of the serial
// step); Walker ~16 k (61.6 us at 16384 against 60.7 us)
static bool in_group_regime(const gx_engine* e)
{
    const int limit = e->cfg.robot == AntRobot::kId ? 27000 : (e->cfg.robot == WalkerRobot::kId ? 16000 : 16384);
    return e->p.N <= limit;
}

// floats of the dynamics tape of a T-step rollout, rounded up so that what follows it in a shard buffer (the layout
// snapshot, float4 planes) stays 16-byte aligned: a tape row is 10 floats for the Point
static size_t tape_floats_padded(const gx_engine* e, int32_t T)
{
    return ((size_t)T * e->p.N * split_tape_width(e->p) + 3) / 4 * 4;
}

// fused rollouts at small env_num: two kernels (a serial dynamics tape, then one thread per
// (step, env) row) instead of the persistent lane-group kernel, from 8 steps up
static bool use_split_rollout(const gx_engine* e, int T)
{
    if (e->path_mode == 1 || e->path_mode == 2) return false;
    if (!split_rollout_supported(e->p) || !in_group_regime(e)) return false;
    return e->path_mode == 3 || T >= 8;
}

// Is other work of this engine on the chip while a dynamics pass runs?  A prefetch sampler in flight, or -- the engine
// is a rank of a multi-GPU run (layout_source 1: pools come from the ranks' export blocks) -- the rank's share of the
// sampler and the expansion of the gathered tapes.  The dynamics pass then takes its compact form (Swimmer: one lane per
// env; Ant / Walker: four envs per wave), which leaves SIMDs to the company; alone it takes the spread-out one.  Ant,
// one GPU playing rank 0 of 8, every tape expanded: 1.73 ms per epoch with the spread-out form, 1.49 ms with the compact.
static bool dyn_pass_has_company(const gx_engine* e)
{
    return (e->pf_valid && e->prefetch_steps != -1) || e->layout_source == 1;
}

static bool use_group_path(const gx_engine* e)
{
    if (e->path_mode == 1) return false;
    if (e->path_mode == 2) return true;
    return in_group_regime(e); // latency regime (path_mode 3: rollouts split, steps here)
}

// ---- does a new stream share the hardware queue of the default stream? ---------------------------------------------
// HIP multiplexes a process's streams onto a few hardware queues.  Rounds 3-4 blamed several slow-downs on a sampler /
// hand-off stream that "aliased" the queue the caller steps on.  This test decides it: a kernel that spins for ~0.3 ms goes
// onto the device's default stream, a marker kernel onto the candidate; if the marker completes while the spinner still
// runs, the two are on different queues.  Measured in round 5 (GX_STREAM_CHECK=1 GX_STREAM_CHECK_VERBOSE=1
// tools/probes/many_engines.py: 24 engines created and destroyed in one process, 1-3 sampler streams each plus the aux
// stream, with 0-5 other torch streams alive): NOT ONE least-priority stream ever shared the default stream's queue.  The
// slow-downs were contention (two samplers in flight beside a chain that is the epoch) and a host-bound harness, not
// aliasing.  The test stays available (GX_STREAM_CHECK=1: a candidate that fails is kept alive until the search is over,
// so that the next one gets another queue, six tries); it is OFF by default.
__global__ void gx_spin_kernel(unsigned long long ticks_100mhz)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks_100mhz) __builtin_amdgcn_s_sleep(32);
}
__global__ void gx_mark_kernel() {}

static bool stream_runs_beside_default(hipStream_t s)
{
    hipEvent_t e_spin = nullptr, e_mark = nullptr;
    if (hipEventCreateWithFlags(&e_spin, hipEventDisableTiming) != hipSuccess) return true;
    if (hipEventCreateWithFlags(&e_mark, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(e_spin); return true; }
    hipLaunchKernelGGL(gx_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)0, 30000ull);
    (void)hipEventRecord(e_spin, (hipStream_t)0);
    hipLaunchKernelGGL(gx_mark_kernel, dim3(1), dim3(64), 0, s);
    (void)hipEventRecord(e_mark, s);
    (void)hipEventSynchronize(e_mark);
    const bool beside = hipEventQuery(e_spin) == hipErrorNotReady; // the marker finished while the spinner was still running
    (void)hipEventSynchronize(e_spin);
    (void)hipGetLastError();
    (void)hipEventDestroy(e_spin); (void)hipEventDestroy(e_mark);
    return beside;
}

// Priority of the engine's own streams.  Rounds 3-5 created all of them at the LEAST priority: they carry throughput work
// (the layout sampler, the hand-off's installs and expansions) beside a chain that is the epoch.  Who issues on a shared
// SIMD is decided by the waves' own priority (s_setprio in the chain's kernels), though, and with the tape hand-off's
// streams in flight a queue of another priority class costs far more than it gives on this stack: one GPU playing rank 0
// of 8 (tools/rehearse_rank.py, profiles/r05_ab_stream_priorities.log), everything queued behind the dynamics pass
// started 85-135 us late unless the sampler's AND the hand-off's queue were of the ordinary class -- Ant rank epoch
// 1.57 -> 1.32 ms, Swimmer 0.71 -> 0.48, Point 0.485 -> 0.43 with every rank's tape expanded; bench.py --gpus 1 over a
// one-rank RCCL group (GX_FORCE_DIST=1) 480 -> 644 M env-steps/s.  The one-GPU epochs do not care, but the per-call
// Engine.step loop (host-bound, small launches) loses 6 % to an ordinary-priority sampler.  Hence two classes: the
// sampler runs on a least-priority stream while the engine is on its own, and on an ordinary one (like the hand-off's
// stream) from the moment it works beside a hand-off.  (GX_SIDE_PRIORITY / GX_AUX_PRIORITY = -1 | 0 | 1: least, ordinary,
// highest, for the first / the second class -- experiments.)
static int side_stream_priority(const char* env, int dflt, int lo, int hi)
{
    const char* ev = getenv(env);
    const int v = ev ? atoi(ev) : dflt;
    return v > 0 ? hi : (v == 0 ? (lo + hi) / 2 : lo);
}

static hipError_t create_side_stream(hipStream_t* out, int prio)
{
    // a stream of ordinary priority may share the default stream's queue: checked unless GX_STREAM_CHECK=0; the
    // least-priority ones never did (above): checked only with GX_STREAM_CHECK=1
    static const int check_env = [] { const char* e = getenv("GX_STREAM_CHECK"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
    int lo_ = 0, hi_ = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo_, &hi_);
    const bool check = check_env >= 0 ? check_env == 1 : prio != lo_;
    hipStream_t rejected[6];
    int nrej = 0;
    hipStream_t s = nullptr;
    hipError_t err = hipSuccess;
    for (int attempt = 0; attempt < 6; ++attempt) {
        err = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio);
        if (err != hipSuccess) break;
        if (!check || attempt == 5 || stream_runs_beside_default(s)) break;
        if (getenv("GX_STREAM_CHECK_VERBOSE")) fprintf(stderr, "guardx: stream candidate %d shares the default stream's queue: rejected\n", attempt);
        rejected[nrej++] = s;
        s = nullptr;
    }
    for (int i = 0; i < nrej; ++i) (void)hipStreamDestroy(rejected[i]);
    *out = s;
    return err;
}

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
            changed = (hipSetDevice(dev) == hipSuccess);
        }
    }
    ~DeviceGuard()
    {
        if (changed) (void)hipSetDevice(prev);
    }
};

// smallest float x with fl(sqrtf(x)) >= thr, so that  sqrtf(d2) < thr  <=>  d2 < x  exactly
// (sqrtf is correctly rounded and monotone).  thr <= 0 or NaN: nothing is ever "< thr".
static float sqrt_cutoff(float thr)
{
    if (!(thr > 0.0f)) return 0.0f;
    if (std::isinf(thr)) return thr;
    float c = thr * thr;
    while (c > 0.0f && sqrtf(std::nextafterf(c, 0.0f)) >= thr) c = std::nextafterf(c, 0.0f);
    while (sqrtf(c) < thr) c = std::nextafterf(c, INFINITY);
    return c;
}

extern "C" const char* gx_last_error(void) { return g_err.c_str(); }
extern "C" int32_t gx_abi_version(void) { return 2; }
#ifndef GX_BUILD_ID
#define GX_BUILD_ID "unknown"
#endif
extern "C" const char* gx_build_id(void) { return GX_BUILD_ID; } // guardx_amd/build.py:source_hash() of the sources
#ifndef GX_BUILD_COMPILER
#define GX_BUILD_COMPILER "unknown"
#endif
extern "C" const char* gx_build_compiler(void) { return GX_BUILD_COMPILER; } // `hipcc --version` the library was built with
extern "C" int32_t gx_obs_dim(const gx_engine* e) { return e ? e->p.D : -1; }
extern "C" int32_t gx_act_dim(const gx_engine* e) { return e ? e->na : -1; }
extern "C" gx_status gx_dims(const gx_engine* e, int32_t* nq, int32_t* nv, int32_t* nu, int32_t* na)
{
    if (!e || !nq || !nv || !nu || !na) return fail(GX_ERR_ARG, "null argument");
    *nq = e->nq; *nv = e->nv; *nu = e->nu; *na = e->na;
    return GX_OK;
}

extern "C" gx_status gx_create(const gx_config* cfg, gx_engine** out)
{
    if (!cfg || !out) return fail(GX_ERR_ARG, "null argument");
    if (cfg->struct_size != (int32_t)sizeof(gx_config))
        return fail(GX_ERR_ARG, "gx_config.struct_size mismatch");
    if (cfg->robot < PointRobot::kId || cfg->robot > PointBareRobot::kId)
        return fail(GX_ERR_UNSUPPORTED,
                    "robots with HIP dynamics: 0 = xmls/point.xml, 1 = xmls/swimmer.xml, 2 = xmls/ant.xml, "
                    "3 = xmls/walker.xml, 4 = xmls/point.xml without the actuator class defaults");
    if (cfg->env_num < 1 || cfg->env_total < cfg->env_num || cfg->env_offset < 0 ||
        cfg->env_offset + cfg->env_num > cfg->env_total)
        return fail(GX_ERR_ARG, "bad env_num/env_total/env_offset");
    if (cfg->hazards_num < 1 || cfg->hazards_num > 64)
        return fail(GX_ERR_ARG, "hazards_num must be in [1,64]");
    if (cfg->pillars_num < 0 || cfg->hazards_num + cfg->pillars_num > 64)
        return fail(GX_ERR_ARG, "pillars_num must be >= 0 and hazards_num + pillars_num <= 64");
    if (cfg->lidar_num_bins < 3 || cfg->lidar_num_bins > 64)
        return fail(GX_ERR_ARG, "lidar_num_bins must be in [3,64]");
    if (cfg->n_candidates < 1 || cfg->physics_steps < 1) return fail(GX_ERR_ARG, "bad n_candidates/physics_steps");

    int ndev = 0;
    GX_HIP(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(GX_ERR_ARG, "bad device ordinal");
    DeviceGuard guard(cfg->device);

    gx_engine* e = new (std::nothrow) gx_engine();
    if (!e) return fail(GX_ERR_HIP, "out of host memory");
    e->cfg = *cfg;
    e->device = cfg->device;
    Params& p = e->p;
    p.robot = cfg->robot;
    if (cfg->robot == SwimmerRobot::kId) {
        e->nq = SwimmerRobot::NQ; e->nv = SwimmerRobot::NV; e->nu = SwimmerRobot::NU;
        e->na = SwimmerRobot::NA; e->ndyn = SwimmerRobot::NDYN;
    } else if (cfg->robot == AntRobot::kId) {
        e->nq = AntRobot::NQ; e->nv = AntRobot::NV; e->nu = AntRobot::NU;
        e->na = AntRobot::NA; e->ndyn = AntRobot::NDYN;
    } else if (cfg->robot == WalkerRobot::kId) {
        e->nq = WalkerRobot::NQ; e->nv = WalkerRobot::NV; e->nu = WalkerRobot::NU;
        e->na = WalkerRobot::NA; e->ndyn = WalkerRobot::NDYN;
    } else {
        e->nq = PointRobot::NQ; e->nv = PointRobot::NV; e->nu = PointRobot::NU;
        e->na = PointRobot::NA; e->ndyn = PointRobot::NDYN;
    }
    p.N = cfg->env_num;
    p.Npad = (p.N + 255) / 256 * 256;
    p.H = cfg->hazards_num;
    p.PL = cfg->pillars_num;
    p.nobj = 1 + p.H + p.PL;
    p.P = (p.nobj + 1) / 2;
    p.bins = cfg->lidar_num_bins;
    // flat obs = concat over sorted(obs_space_dict keys)  engine.py:386-409,773-777
    int o = 0;
    p.off_acc = p.off_ctrl = p.off_comp = p.off_gl = p.off_hl = p.off_pl = p.off_qpos = p.off_qvel = p.off_vel = -1;
    if (cfg->observe_acc) { p.off_acc = o; o += 2; }
    if (cfg->observe_ctrl) { p.off_ctrl = o; o += e->nu; }
    if (cfg->observe_goal_comp) { p.off_comp = o; o += 2; }
    if (cfg->observe_goal_lidar) { p.off_gl = o; o += p.bins; }
    if (cfg->observe_hazards) { p.off_hl = o; o += p.bins; }
    if (cfg->observe_pillars && p.PL > 0) { p.off_pl = o; o += p.bins; } // 'pillars_lidar' sorts between hazards_lidar and qpos
    if (cfg->observe_qpos) { p.off_qpos = o; o += e->nq; }
    if (cfg->observe_qvel) { p.off_qvel = o; o += e->nv; }
    if (cfg->observe_vel) { p.off_vel = o; o += 2; }
    p.D = o;
    if (p.D < 1) { delete e; return fail(GX_ERR_ARG, "empty observation"); }
    p.lidar_alias = cfg->lidar_alias;
    p.lidar_max_dist_set = cfg->lidar_max_dist_set;
    p.lidar_max_dist = cfg->lidar_max_dist;
    p.neg_gain = -cfg->lidar_exp_gain;
    p.bin_size = (float)((3.14159265358979323846 * 2) / p.bins); // engine.py:880
    p.goal_size = cfg->goal_size;
    p.goal_cut = sqrt_cutoff(cfg->goal_size);
    {   // robot_rot: rot2quat -> (w, 0, 0, z); the body's x axis in the world is (w^2 - z^2, 2 w z) in fp32
        const float w = (float)cos(0.5 * (double)cfg->robot_rot), z = (float)sin(0.5 * (double)cfg->robot_rot);
        p.rot_on = cfg->robot_rot != 0.0f;
        p.rot_c = w * w - z * z;
        p.rot_s = 2.0f * (w * z);
    }
    p.hazards_size = cfg->hazards_size;
    p.pillars_size = cfg->pillars_size;
    p.reward_distance = cfg->reward_distance;
    p.num_steps_f = (float)cfg->num_steps;
    p.physics_steps = cfg->physics_steps;
    const float h_robot = cfg->robot == SwimmerRobot::kId ? SwimmerRobot::kH
                          : cfg->robot == AntRobot::kId   ? AntRobot::kH
                          : cfg->robot == WalkerRobot::kId ? WalkerRobot::kH
                                                          : PointRobot::kH;
    p.dt = h_robot * (float)cfg->physics_steps; // engine.py:235
    p.env_total = cfg->env_total;
    p.env_offset = cfg->env_offset;
    p.have_last = p.have_last_last = 0;
    p.hist_on = (cfg->observe_vel || cfg->observe_acc) ? 1 : 0;

    // layout sampler constants (python-float arithmetic, then f32: engine.py:554,574-577)
    SampleParams& sp = e->sp;
    e->nobj_total = p.H + p.PL + 2;
    sp.M = cfg->n_candidates;
    sp.c0 = 0; sp.Mtot = sp.M;
    sp.dbg = nullptr;
    sp.nobj_total = e->nobj_total;
    sp.H = p.H;
    const double ko[4] = {cfg->goal_keepout, cfg->hazards_keepout, cfg->robot_keepout, cfg->pillars_keepout};
    std::vector<float4> hb; // per-hazard / per-pillar rectangles when explicit placements are given
    for (int t = 0; t < 4; ++t) {
        // object of this type whose rectangle seeds the per-type bounds: goal = 0, robot = last
        const double* rc = cfg->extents;
        if (cfg->placements && t == 0) rc = &cfg->placements[0];
        if (cfg->placements && t == 2) rc = &cfg->placements[4 * (p.H + p.PL + 1)];
        sp.lo_x[t] = (float)(rc[0] + ko[t]);
        sp.lo_y[t] = (float)(rc[1] + ko[t]);
        sp.hi_x[t] = (float)(rc[2] - ko[t]);
        sp.hi_y[t] = (float)(rc[3] - ko[t]);
        for (int q = 0; q < 4; ++q) sp.thr[q][t] = (float)(ko[q] + cfg->placements_margin + ko[t]);
    }
    sp.haz_bounds = nullptr;
    if (cfg->placements)
        for (int hz = 0; hz < p.H + p.PL; ++hz) {
            const double* rc = &cfg->placements[4 * (1 + hz)];
            const double k = hz < p.H ? ko[1] : ko[3];
            hb.push_back(make_float4((float)(rc[0] + k), (float)(rc[2] - k), (float)(rc[1] + k), (float)(rc[3] - k)));
        }
    sp.min_rg = cfg->robot_goal_min_dist;
    for (int q = 0; q < 4; ++q)
        for (int t = 0; t < 4; ++t) sp.thr_sq[q][t] = sqrt_cutoff(sp.thr[q][t]);
    sp.min_rg_sq = sqrt_cutoff(sp.min_rg);
    // Which form of the sampler (gx_kernels.hip, sample_phase2_kernel<kFused>)?  Phase 1 pays a whole walk of the key chain
    // to reject the candidates none of whose ten robot tries can be min_rg away from the goal.  Its pass rate depends on
    // the geometry only -- goal uniform in its rectangle, ten robot tries uniform in theirs -- and is estimated here by a
    // fixed-seed Monte Carlo (a plain LCG: an estimate for a cost decision, not part of the sampler; both forms give the
    // same pool).  Reference arena (4 m, min_rg 3.0): ~0.25 -> three phases; synthetic config 5 (6 m): ~0.9 -> fused.
    {
        uint64_t lcg = 0x9E3779B97F4A7C15ull;
        auto u01 = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (double)(lcg >> 11) * (1.0 / 9007199254740992.0); };
        int pass = 0;
        const int trials = 4000;
        for (int i = 0; i < trials; ++i) {
            const double gx = sp.lo_x[0] + u01() * (sp.hi_x[0] - sp.lo_x[0]), gy = sp.lo_y[0] + u01() * (sp.hi_y[0] - sp.lo_y[0]);
            bool far = false;
            for (int t = 0; t < 10; ++t) {
                const double rx = sp.lo_x[2] + u01() * (sp.hi_x[2] - sp.lo_x[2]), ry = sp.lo_y[2] + u01() * (sp.hi_y[2] - sp.lo_y[2]);
                if ((rx - gx) * (rx - gx) + (ry - gy) * (ry - gy) >= (double)sp.min_rg * sp.min_rg) far = true;
            }
            pass += far ? 1 : 0;
        }
        e->phase1_pass_estimate = (float)pass / trials;
        sp.fused = e->phase1_pass_estimate > 0.6f ? 1 : 0;
        if (const char* ev = getenv("GX_SAMPLE_FUSED")) sp.fused = atoi(ev) ? 1 : 0; // experiments / tests
    }

    // PRNGKey(seed)  engine.py:216
    e->key[0] = 0u;
    e->key[1] = cfg->seed;
    e->hist = 0;
    e->have_reset = false;
    e->layout_pending = false;
    e->h_layout_size = nullptr;
    e->path_mode = 0;
    e->policy_steps = 0;
    e->policy_impl = 0;
    e->haz_bounds = nullptr;
    e->cfg.placements = nullptr; // not retained (folded into SampleParams above)
    e->pf_valid = false;
    e->prefetch_steps = -2;
    e->steps_since_reset = 0; e->last_interval = 0; e->pf_hits = 0; e->pf_misses = 0;
    for (int i = 0; i < gx_engine::kPools; ++i) e->side[i] = nullptr;
    e->n_side = 1;
    memset(e->pools, 0, sizeof(e->pools));
    for (int i = 0; i < gx_engine::kPools; ++i) {
        e->pool_ready[i] = nullptr; e->pool_free[i] = nullptr; e->expand_ev[i] = nullptr;
        e->expand_pending[i] = false; e->pool_gen[i] = 0;
    }
    memset(&e->b, 0, sizeof(e->b));
    memset(&e->shard_scratch, 0, sizeof(e->shard_scratch));
    memset(e->jobs, 0, sizeof(e->jobs));

    const size_t M = (size_t)sp.M;
    hipError_t err = hipSuccess;
    auto alloc = [&](void** ptr, size_t bytes) {
        if (err == hipSuccess) err = hipMalloc(ptr, bytes);
        if (err == hipSuccess) err = hipMemset(*ptr, 0, bytes);
    };
    alloc((void**)&e->b.dyn, sizeof(float4) * e->ndyn * p.Npad);
    alloc((void**)&e->b.obj, sizeof(float4) * (size_t)p.P * p.Npad);
    alloc((void**)&e->b.hist, sizeof(float4) * p.Npad);
    alloc((void**)&e->b.rd_j, sizeof(int) * p.Npad);
    if (err == hipSuccess) err = hipMemset(e->b.rd_j, 0xFF, sizeof(int) * p.Npad); // -1: nothing speculated
    if (!hb.empty()) {
        alloc((void**)&e->haz_bounds, sizeof(float4) * hb.size());
        if (err == hipSuccess) err = hipMemcpy(e->haz_bounds, hb.data(), sizeof(float4) * hb.size(), hipMemcpyHostToDevice);
        sp.haz_bounds = e->haz_bounds;
    }
    for (int i = 0; i < gx_engine::kPools; ++i) {
        Pool& pl = e->pools[i];
        const size_t tile = (size_t)sample_compact_tile();
        alloc((void**)&pl.cand_ok, (M + tile - 1) / tile * tile);
        alloc((void**)&pl.cand_xy, sizeof(float2) * M * e->nobj_total);
        alloc((void**)&pl.blk_cnt, sizeof(int) * ((M + tile - 1) / tile));
        alloc((void**)&pl.cand_of, sizeof(int) * M);
        alloc((void**)&pl.layout_size, sizeof(int));
        alloc((void**)&pl.n_surv, 2 * sizeof(int));
        alloc((void**)&pl.surv, sizeof(uint32_t) * 32 * M);
        alloc((void**)&pl.surv0, sizeof(uint32_t) * 8 * M);
        pl.fake = nullptr;
        if (fake_table_width(p) > 0) alloc((void**)&pl.fake, sizeof(float) * (size_t)fake_table_width(p) * M);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&e->pool_ready[i], hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&e->pool_free[i], hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&e->expand_ev[i], hipEventDisableTiming);
    }
    if (err == hipSuccess) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi); // lo = least urgent
        int prio = side_stream_priority("GX_SIDE_PRIORITY", -1, lo, hi);
        // ONE side stream by default.  One per pool lets consecutive samplers overlap their tails (+1.8 % on the
        // headline epoch), but HIP multiplexes streams onto 4 hardware queues: with the caller's stream, the tape
        // hand-off's expansion stream and RCCL's own stream -- or a second engine in the process -- a fourth and
        // fifth stream alias, and a sampler queued behind the stepping it should overlap costs 25 % (bench.py's
        // extras, which keep the headline engine alive, showed exactly the serial sum).
        e->n_side = 1;
        if (const char* ev = getenv("GX_SIDE_STREAMS")) e->n_side = atoi(ev) >= 1 && atoi(ev) <= gx_engine::kPools ? atoi(ev) : e->n_side; // experiments
        for (int i = 0; i < e->n_side && err == hipSuccess; ++i)
            err = create_side_stream(&e->side[i], prio);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&e->side_switch, hipEventDisableTiming);
    }
    e->cur = 0;
    e->b.pool = e->pools[0];
    if (err == hipSuccess) err = hipHostMalloc((void**)&e->h_layout_size, 2 * sizeof(int), hipHostMallocMapped);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&e->layout_ev, hipEventDisableTiming);
    if (err == hipSuccess) {
        // xmat of the zero pose: cos = 1 (engine.py:229 MjData default); it sits in .x of the last
        // float4 of the dynamic state for every robot
        std::vector<float4> dl(p.Npad, make_float4(1.f, 0.f, 0.f, 0.f));
        err = hipMemcpy(e->b.dyn + (size_t)(e->ndyn - 1) * p.Npad, dl.data(), sizeof(float4) * p.Npad,
                        hipMemcpyHostToDevice);
    }
    if (err != hipSuccess) {
        std::string m = std::string("device allocation failed: ") + hipGetErrorString(err);
        gx_destroy(e);
        return fail(GX_ERR_HIP, m);
    }
    e->h_layout_size[0] = 0;
    e->h_layout_size[1] = 0x7fffffff;
    *out = e;
    return GX_OK;
}

static void free_shard_scratch(gx_engine* e)
{
    Pool& pl = e->shard_scratch;
    void* pb[] = {pl.cand_ok, pl.cand_xy, pl.blk_cnt, pl.cand_of, pl.layout_size, pl.n_surv, pl.surv, pl.surv0};
    for (void* q : pb)
        if (q) (void)hipFree(q);
    memset(&pl, 0, sizeof(pl));
    e->shard_scratch_ok = false;
    e->shard_scratch_cap = 0;
}

extern "C" gx_status gx_destroy(gx_engine* e)
{
    if (!e) return GX_OK;
    DeviceGuard guard(e->device);
    (void)hipDeviceSynchronize();
    void* bufs[] = {e->b.dyn, e->b.obj, e->b.hist, e->b.rd_j, e->haz_bounds, e->tape, e->obj0, e->pol_wt, e->pol_cur};
    for (void* q : bufs)
        if (q) (void)hipFree(q);
    for (int i = 0; i < gx_engine::kPools; ++i) {
        Pool& pl = e->pools[i];
        void* pb[] = {pl.cand_ok, pl.cand_xy, pl.blk_cnt, pl.cand_of, pl.layout_size, pl.n_surv, pl.surv, pl.surv0, pl.fake};
        for (void* q : pb)
            if (q) (void)hipFree(q);
        if (e->pool_ready[i]) (void)hipEventDestroy(e->pool_ready[i]);
        if (e->pool_free[i]) (void)hipEventDestroy(e->pool_free[i]);
        if (e->expand_ev[i]) (void)hipEventDestroy(e->expand_ev[i]);
    }
    for (int i = 0; i < gx_engine::kPools; ++i)
        if (e->side[i]) (void)hipStreamDestroy(e->side[i]);
    if (e->side_ord) (void)hipStreamDestroy(e->side_ord);
    if (e->side_switch) (void)hipEventDestroy(e->side_switch);
    if (e->h_keys) (void)hipHostFree(e->h_keys);
    if (e->h_layout_size) (void)hipHostFree(e->h_layout_size);
    if (e->layout_ev) (void)hipEventDestroy(e->layout_ev);
    if (e->pf_phase1) (void)hipEventDestroy(e->pf_phase1);
    if (e->shard_dep) (void)hipEventDestroy(e->shard_dep);
    if (e->shard_done) (void)hipEventDestroy(e->shard_done);
    free_shard_scratch(e);
    delete e;
    return GX_OK;
}

// install a requested-but-not-yet-installed reset_done before a consumer that does not apply it on load
static gx_status flush_pending(gx_engine* e, hipStream_t s)
{
    if (take_commit(e)) {
        launch_commit_pending(e->p, e->b, e->nobj_total, e->sp.M, s);
        GX_HIP(hipGetLastError());
    }
    return GX_OK;
}

// a sampler is about to overwrite pool i on stream `s`: tape tokens of that pool expire, and the sampler runs
// behind the last gx_expand_tape that reads the pool
static hipError_t claim_pool(gx_engine* e, int i, hipStream_t s)
{
    e->pool_gen[i]++;
    if (!e->expand_pending[i]) return hipSuccess;
    e->expand_pending[i] = false;
    return hipStreamWaitEvent(s, e->expand_ev[i], 0);
}

static void layout_keys(const gx_engine* e, uint32_t (&k)[4])
{
    // get_layout: randint(key, ...) splits the key once  engine.py:447
    split2(e->key[0], e->key[1], k[0], k[1], k[2], k[3]);
}

// The second-class sampler stream exists from the moment the engine is told about a hand-off (gx_aux_stream: before the
// caller creates its collective's streams) -- an engine on its own keeps the streams, and hardware queues, it always had.
// HIP maps a process's streams of one priority onto a few hardware queues in creation order; created later (at the first
// sampler launch, after the rehearsal's copy stream) this stream shared that stream's queue and the sampler queued behind
// the "collective" it should overlap (profiles/r05_ab_stream_priorities.log).
static hipError_t ensure_side_ord(gx_engine* e)
{
    if (e->side_ord) return hipSuccess;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    return create_side_stream(&e->side_ord, side_stream_priority("GX_AUX_PRIORITY", 0, lo, hi));
}

// the stream the next layout sampler goes onto; a change of class is ordered behind the samplers of the other class (they
// share scratch arrays and the pool ring)
static hipError_t sampler_stream(gx_engine* e, int pool, hipStream_t* out)
{
    const int cls = (e->aux_in_use || e->layout_source == 1) ? 1 : 0;
    if (cls) {
        const hipError_t err = ensure_side_ord(e);
        if (err != hipSuccess) return err;
    }
    hipStream_t want = cls ? e->side_ord : e->side[pool % e->n_side];
    if (cls != e->sampler_class) {
        hipError_t err = hipSuccess;
        if (cls) {
            for (int i = 0; i < e->n_side && err == hipSuccess; ++i) {
                err = hipEventRecord(e->side_switch, e->side[i]);
                if (err == hipSuccess) err = hipStreamWaitEvent(e->side_ord, e->side_switch, 0);
            }
        } else {
            err = hipEventRecord(e->side_switch, e->side_ord);
            for (int i = 0; i < e->n_side && err == hipSuccess; ++i) err = hipStreamWaitEvent(e->side[i], e->side_switch, 0);
        }
        if (err != hipSuccess) return err;
        e->sampler_class = cls;
    }
    *out = want;
    return hipSuccess;
}

extern "C" gx_status gx_reset(gx_engine* e, float* d_obs, void* stream)
{
    if (!e || !d_obs) return fail(GX_ERR_ARG, "null argument");
    DeviceGuard guard(e->device);
    hipStream_t s = (hipStream_t)stream;
    (void)take_commit(e); // reset() re-initialises every env: a pending reset_done is moot (engine.py:460-465)
    const int other = (e->cur + 1) % gx_engine::kPools;
    const bool hit = e->pf_valid && e->pf_key[0] == e->key[0] && e->pf_key[1] == e->key[1];
    if (e->pf_valid) { if (hit) e->pf_hits++; else e->pf_misses++; }
    if (e->have_reset) e->last_interval = e->steps_since_reset;
    e->steps_since_reset = 0;
    const bool swap = hit || e->have_reset;
    // the old pool is free once everything ALREADY queued on `s` has run: recorded before `s` starts to wait
    // for the side stream, so the next prefetch (which reuses the old pool) follows the current one
    // back to back instead of one cross-stream round trip later
    if (swap) GX_HIP(hipEventRecord(e->pool_free[e->cur], s));
    if (e->pf_valid) // whatever the side stream is doing to pools[other] finishes first
        GX_HIP(hipStreamWaitEvent(s, e->pool_ready[other], 0));
    if (swap) e->cur = other;
    if (!hit) { // reset_layout on the caller's stream  engine.py:433-444
        GX_HIP(claim_pool(e, e->cur, s));
        e->sp.k0 = e->key[0];
        e->sp.k1 = e->key[1];
        // per-wave stamps of sample_phase2 (tools/history/debug/sampler_waves.py): only on request, the buffer must hold
        // 65536 + 4 * 16384 words -- the other stamp tools pass much smaller ones
        e->sp.dbg = (e->stamps && getenv("GX_SAMPLER_STAMPS")) ? e->stamps + 65536 : nullptr;
        GX_HIP(launch_sample(e->sp, e->pools[e->cur], s));
        launch_fake_table(e->p, e->pools[e->cur], e->nobj_total, e->sp.M, s);
    }
    e->pf_valid = false;
    e->b.pool = e->pools[e->cur];
    uint32_t k[4];
    layout_keys(e, k);
    launch_reset_apply(e->p, e->b, e->nobj_total, k[0], k[1], k[2], k[3], d_obs, e->h_layout_size, s);
    GX_HIP(hipEventRecord(e->layout_ev, s));
    GX_HIP(hipGetLastError());
    e->layout_pending = true;
    const bool first = !e->have_reset;
    e->have_reset = true;

    // prefetch the pool of the next reset(): the key then is this key advanced by one split per
    // step() (engine.py:431) -- independent of the data, so it can be computed now
    const int horizon = e->prefetch_steps == -2 ? (e->last_interval > 0 ? e->last_interval : e->cfg.num_steps)
                                                 : e->prefetch_steps;
    if (horizon >= 0 && e->layout_source == 0) {
        uint32_t k0 = e->key[0], k1 = e->key[1];
        for (int t = 0; t < horizon; ++t) {
            uint32_t a0, a1, b0, b1;
            split2(k0, k1, a0, a1, b0, b1);
            k0 = a0; k1 = a1;
        }
        const int tgt = (e->cur + 1) % gx_engine::kPools;
        hipStream_t side = nullptr;
        GX_HIP(sampler_stream(e, tgt, &side));
        if (!first) GX_HIP(hipStreamWaitEvent(side, e->pool_free[tgt], 0));
        GX_HIP(claim_pool(e, tgt, side));
        // The closed-loop policy kernel (256-thread workgroups holding ~150 KB of LDS each) loses ~15 % when the
        // sampler's grids reach the CUs first; after a policy rollout the prefetch therefore starts behind
        // reset_apply.  The open-loop kernels are insensitive and keep the back-to-back sampler chain.
        if (e->last_policy) GX_HIP(hipStreamWaitEvent(side, e->layout_ev, 0));
        SampleParams sp = e->sp;
        sp.k0 = k0; sp.k1 = k1;
        if (!e->pf_phase1) GX_HIP(hipEventCreateWithFlags(&e->pf_phase1, hipEventDisableTiming));
        GX_HIP(launch_sample(sp, e->pools[tgt], side, e->pf_phase1));
        launch_fake_table(e->p, e->pools[tgt], e->nobj_total, sp.M, side);
        e->pf_phase1_pending = true;
        GX_HIP(hipEventRecord(e->pool_ready[tgt], side));
        GX_HIP(hipGetLastError());
        e->pf_valid = true;
        e->pf_key[0] = k0; e->pf_key[1] = k1;
    }
    return GX_OK;
}

// ---- sharded layout sampling (optional, multi-GPU) -------------------------------------------------------------
// The candidates of reset()'s 1e6-candidate rejection sampler are independent (candidate c uses split(key, 1e6)[c]), so
// rank r of W can sample candidates [r M / W, (r + 1) M / W) alone, export its valid layouts in candidate order, and
// after ONE all-gather of the exports (a few MB) every rank installs the same pool the unsharded sampler compacts:
// the 0.5 ms that bound the epoch are done once per node instead of once per GPU.  A second collective on a path whose
// north_star allows one (the rollout hand-off): off unless the caller asks for it (guardx_amd.dist.ShardedReset).
static int shard_target_pool(const gx_engine* e) { return e->have_reset ? (e->cur + 1) % gx_engine::kPools : e->cur; }

extern "C" gx_status gx_sample_shard(gx_engine* e, int32_t shard, int32_t n_shards, float* d_rows, int32_t cap,
                                     int32_t* d_count, void* stream)
{
    if (!e || !d_rows || !d_count || n_shards < 1 || shard < 0 || shard >= n_shards || cap < 1)
        return fail(GX_ERR_ARG, "gx_sample_shard: bad argument");
    if (e->pf_valid || e->prefetch_steps != -1)
        return fail(GX_ERR_STATE, "gx_sample_shard: switch the layout prefetch off first (gx_set_prefetch(e, -1) before the "
                                  "reset in front of this one): a sharded reset samples on the caller's stream");
    DeviceGuard guard(e->device);
    hipStream_t s = (hipStream_t)stream;
    const int tgt = shard_target_pool(e);
    GX_HIP(claim_pool(e, tgt, s));
    e->rs_sampled = true; e->rs_pool = tgt; e->rs_key[0] = e->key[0]; e->rs_key[1] = e->key[1];
    SampleParams sp = e->sp;
    const long long M = e->sp.M;
    sp.c0 = (int)(M * shard / n_shards);
    sp.M = (int)(M * (shard + 1) / n_shards) - sp.c0;
    sp.Mtot = (int)M;
    sp.k0 = e->key[0]; sp.k1 = e->key[1];
    sp.dbg = nullptr;
    if (sp.M < 1) return fail(GX_ERR_ARG, "gx_sample_shard: more shards than candidates");
    // the flags behind this shard's last candidate, up to the end of its last compaction tile, may be a full-size
    // sampler's: the ordered compaction reads whole tiles
    const int tile = sample_compact_tile();
    const int padded = (sp.M + tile - 1) / tile * tile;
    if (padded > sp.M) GX_HIP(hipMemsetAsync(e->pools[tgt].cand_ok + sp.M, 0, (size_t)(padded - sp.M), s));
    GX_HIP(launch_sample(sp, e->pools[tgt], s));
    launch_pool_export(e->pools[tgt], e->nobj_total, reinterpret_cast<float2*>(d_rows), cap, d_count, s);
    GX_HIP(hipGetLastError());
    return GX_OK;
}

extern "C" gx_status gx_reset_from_shards(gx_engine* e, const float* d_rows_all, const int32_t* d_counts, int32_t n_shards,
                                          int32_t cap, float* d_obs, void* stream)
{
    if (!e || !d_rows_all || !d_counts || !d_obs || n_shards < 1 || cap < 1)
        return fail(GX_ERR_ARG, "gx_reset_from_shards: bad argument");
    if (e->pf_valid || e->prefetch_steps != -1) return fail(GX_ERR_STATE, "gx_reset_from_shards: layout prefetch is on");
    DeviceGuard guard(e->device);
    hipStream_t s = (hipStream_t)stream;
    (void)take_commit(e);
    if (e->have_reset) e->last_interval = e->steps_since_reset;
    e->steps_since_reset = 0;
    const int tgt = shard_target_pool(e);
    // normally this engine sampled one of the shards itself (gx_sample_shard claimed the pool: tokens of its old
    // contents expired, the sampler ran behind the last expansion that read them).  If it did not -- or sampled for
    // another key -- the pool is claimed here, so that an outstanding tape token never names rewritten rows
    if (!(e->rs_sampled && e->rs_pool == tgt && e->rs_key[0] == e->key[0] && e->rs_key[1] == e->key[1]))
        GX_HIP(claim_pool(e, tgt, s));
    e->rs_sampled = false;
    if (e->have_reset) GX_HIP(hipEventRecord(e->pool_free[e->cur], s));
    e->cur = tgt;
    launch_pool_install(e->pools[tgt], e->nobj_total, n_shards, cap, reinterpret_cast<const float2*>(d_rows_all), d_counts,
                        e->sp.M, s);
    launch_fake_table(e->p, e->pools[tgt], e->nobj_total, e->sp.M, s);
    GX_HIP(hipEventRecord(e->pool_ready[tgt], s)); // gx_expand_tape on another stream waits for the pool it reads
    e->b.pool = e->pools[tgt];
    uint32_t k[4];
    layout_keys(e, k);
    launch_reset_apply(e->p, e->b, e->nobj_total, k[0], k[1], k[2], k[3], d_obs, e->h_layout_size, s);
    GX_HIP(hipEventRecord(e->layout_ev, s));
    GX_HIP(hipGetLastError());
    e->layout_pending = true;
    e->have_reset = true;
    return GX_OK;
}

// ---- piggy-backed form: the shard of a LATER reset travels with the rollout hand-off ---------------------------------
// The key of a later reset is known now (this key advanced by one split per step(), engine.py:431, the horizon being the
// learned interval between resets).  The schedule (include/guardx.h spells it out; guardx_amd/dist.py:TapeHandoff runs
// it): after epoch k's rollout rank r samples ITS share of the candidates of reset(k + 3) on the side stream
// (resets_ahead = 3: the key advanced by 3 * horizon - steps_since_reset) into the tail of the buffer that carries epoch
// k + 1's tape; that epoch's all-gather delivers every rank's block; during epoch k + 2 -- after gx_reset(k + 2), before
// gx_reset(k + 3) -- gx_install_shards turns the blocks into the pool slot the NEXT gx_reset takes as a prefetch hit.
// A reset whose key has no installed pool samples inline (all candidates), exactly as a prefetch miss does: results
// never depend on any of this.
extern "C" gx_status gx_set_layout_source(gx_engine* e, int32_t source)
{
    if (!e || source < 0 || source > 1) return fail(GX_ERR_ARG, "gx_set_layout_source: 0 = own sampler, 1 = installed shards");
    e->layout_source = source;
    return GX_OK;
}

static int shard_horizon(const gx_engine* e)
{
    return e->prefetch_steps == -2 ? (e->last_interval > 0 ? e->last_interval : e->cfg.num_steps) : e->prefetch_steps;
}

extern "C" gx_status gx_shard_block_floats(const gx_engine* e, int32_t cap, int64_t* floats)
{
    if (!e || cap < 1 || !floats) return fail(GX_ERR_ARG, "gx_shard_block_floats: bad argument");
    *floats = 4 + (int64_t)cap * e->nobj_total * 2; // count, key, tag | rows: a multiple of 4 floats
    return GX_OK;
}

extern "C" gx_status gx_sample_shard_ahead(gx_engine* e, int32_t shard, int32_t n_shards, int32_t resets_ahead,
                                           float* d_block, int32_t cap, int64_t* ticket, void* stream)
{
    if (!e || !d_block || !ticket || n_shards < 1 || shard < 0 || shard >= n_shards || cap < 1 || resets_ahead < 1 || n_shards > 32767)
        return fail(GX_ERR_ARG, "gx_sample_shard_ahead: bad argument");
    if (reinterpret_cast<uintptr_t>(d_block) & 15u) return fail(GX_ERR_ARG, "gx_sample_shard_ahead: d_block must be 16-byte aligned");
    if (!e->have_reset) return fail(GX_ERR_STATE, "gx_sample_shard_ahead before gx_reset");
    if (e->layout_source != 1)
        return fail(GX_ERR_STATE, "gx_sample_shard_ahead: gx_set_layout_source(e, 1) first (the engine's own prefetch "
                                  "sampler and installed shards would fight over the next pool)");
    const int horizon = shard_horizon(e);
    if (horizon < 1) return fail(GX_ERR_STATE, "gx_sample_shard_ahead: needs a reset interval (gx_set_prefetch(e, -2) or a fixed one)");
    const long long adv = (long long)resets_ahead * horizon - e->steps_since_reset;
    if (adv < 0 || adv > (1 << 24)) return fail(GX_ERR_STATE, "gx_sample_shard_ahead: more steps since the last reset than the horizon covers");
    DeviceGuard guard(e->device);
    hipStream_t s = (hipStream_t)stream;
    const long long M = e->sp.M;
    SampleParams sp = e->sp;
    sp.c0 = (int)(M * shard / n_shards);
    sp.M = (int)(M * (shard + 1) / n_shards) - sp.c0;
    sp.Mtot = (int)M;
    sp.dbg = nullptr;
    if (sp.M < 1) return fail(GX_ERR_ARG, "gx_sample_shard_ahead: more shards than candidates");
    uint32_t k0 = e->key[0], k1 = e->key[1];
    for (long long t = 0; t < adv; ++t) {
        uint32_t a0, a1, b0, b1;
        split2(k0, k1, a0, a1, b0, b1);
        k0 = a0; k1 = a1;
    }
    sp.k0 = k0; sp.k1 = k1;
    Pool& pl = e->shard_scratch;
    if (!e->shard_scratch_ok || e->shard_scratch_cap < sp.M) {
        // sized for THIS shard's candidates (1/n_shards of the list; ~35 MB instead of ~280 MB at 1e6 / 8), regrown when a
        // later call samples a larger share (n_shards shrank)
        if (e->shard_scratch_ok) {
            GX_HIP(hipStreamSynchronize(e->sampler_class ? e->side_ord : e->side[0])); // the last shard sampler still runs on the old arrays
            free_shard_scratch(e);
        }
        hipError_t err = hipSuccess;
        auto alloc = [&](void** ptr, size_t bytes) {
            if (err == hipSuccess) err = hipMalloc(ptr, bytes);
            if (err == hipSuccess) err = hipMemset(*ptr, 0, bytes);
        };
        const size_t Mz = (size_t)sp.M, tile = (size_t)sample_compact_tile();
        alloc((void**)&pl.cand_ok, (Mz + tile - 1) / tile * tile);
        alloc((void**)&pl.cand_xy, sizeof(float2) * Mz * e->nobj_total);
        alloc((void**)&pl.blk_cnt, sizeof(int) * ((Mz + tile - 1) / tile));
        alloc((void**)&pl.cand_of, sizeof(int) * Mz);
        alloc((void**)&pl.layout_size, sizeof(int));
        alloc((void**)&pl.n_surv, 2 * sizeof(int));
        alloc((void**)&pl.surv, sizeof(uint32_t) * 32 * Mz);
        alloc((void**)&pl.surv0, sizeof(uint32_t) * 8 * Mz);
        pl.fake = nullptr;
        if (err == hipSuccess && !e->shard_dep) err = hipEventCreateWithFlags(&e->shard_dep, hipEventDisableTiming);
        if (err == hipSuccess && !e->shard_done) err = hipEventCreateWithFlags(&e->shard_done, hipEventDisableTiming);
        if (err != hipSuccess) {
            free_shard_scratch(e); // whatever was allocated before the failure
            return fail(GX_ERR_HIP, std::string("gx_sample_shard_ahead: ") + hipGetErrorString(err));
        }
        e->shard_scratch_ok = true;
        e->shard_scratch_cap = sp.M;
    }
    hipStream_t side = nullptr;
    GX_HIP(sampler_stream(e, 0, &side));
    // behind everything queued on the caller's stream: the block may be the tail of a buffer an earlier collective
    // read, and the caller ordered its own stream behind that
    GX_HIP(hipEventRecord(e->shard_dep, s));
    GX_HIP(hipStreamWaitEvent(side, e->shard_dep, 0));
    const int tile = sample_compact_tile();
    const int padded = (sp.M + tile - 1) / tile * tile;
    if (padded > sp.M) GX_HIP(hipMemsetAsync(pl.cand_ok + sp.M, 0, (size_t)(padded - sp.M), side));
    GX_HIP(launch_sample(sp, pl, side));
    launch_pool_export(pl, e->nobj_total, reinterpret_cast<float2*>(d_block + 4), cap, reinterpret_cast<int*>(d_block), side,
                       reinterpret_cast<uint32_t*>(d_block), k0, k1, (uint32_t)shard | ((uint32_t)n_shards << 16));
    GX_HIP(hipEventRecord(e->shard_done, side));
    GX_HIP(hipGetLastError());
    e->shard_inflight = true;
    gx_engine::ShardJob& j = e->jobs[e->next_ticket % gx_engine::kJobs];
    j.ticket = e->next_ticket; j.k0 = k0; j.k1 = k1; j.n_shards = n_shards; j.cap = cap;
    *ticket = e->next_ticket++;
    return GX_OK;
}

// `stream` waits for the export block of the last gx_sample_shard_ahead (call it before handing the block to a collective)
extern "C" gx_status gx_shard_join(gx_engine* e, void* stream)
{
    if (!e) return fail(GX_ERR_ARG, "null engine");
    if (!e->shard_inflight) return GX_OK;
    DeviceGuard guard(e->device);
    GX_HIP(hipStreamWaitEvent((hipStream_t)stream, e->shard_done, 0));
    return GX_OK;
}

extern "C" gx_status gx_install_shards(gx_engine* e, int64_t ticket, const float* d_blocks, int64_t stride_floats,
                                       int32_t n_shards, int32_t cap, void* stream)
{
    if (!e || !d_blocks || n_shards < 1 || cap < 1 || stride_floats < 4 + (int64_t)cap * (e ? e->nobj_total : 0) * 2)
        return fail(GX_ERR_ARG, "gx_install_shards: bad argument");
    if ((reinterpret_cast<uintptr_t>(d_blocks) & 15u) || (stride_floats & 3))
        return fail(GX_ERR_ARG, "gx_install_shards: blocks must be 16-byte aligned and a multiple of 4 floats apart");
    if (!e->have_reset) return fail(GX_ERR_STATE, "gx_install_shards before gx_reset");
    if (e->layout_source != 1) return fail(GX_ERR_STATE, "gx_install_shards: gx_set_layout_source(e, 1) first");
    const gx_engine::ShardJob j = e->jobs[(ticket > 0 ? ticket : 0) % gx_engine::kJobs];
    if (ticket < 1 || j.ticket != ticket)
        return fail(GX_ERR_STATE, "gx_install_shards: unknown ticket (the engine remembers its last four gx_sample_shard_ahead calls)");
    if (j.n_shards != n_shards || j.cap != cap)
        return fail(GX_ERR_STATE, "gx_install_shards: n_shards / cap differ from the gx_sample_shard_ahead call of this ticket");
    DeviceGuard guard(e->device);
    hipStream_t s = (hipStream_t)stream;
    const int tgt = (e->cur + 1) % gx_engine::kPools;
    GX_HIP(hipStreamWaitEvent(s, e->pool_free[tgt], 0)); // the epoch that drew from this pool has been stepped
    if (e->pf_valid) GX_HIP(hipStreamWaitEvent(s, e->pool_ready[tgt], 0)); // a sampler / install still writing it finishes first
    GX_HIP(claim_pool(e, tgt, s));                       // tokens of its old rows expire; behind their last expansion
    launch_pool_install_blocks(e->pools[tgt], e->nobj_total, n_shards, cap, d_blocks, stride_floats, j.k0, j.k1, e->sp.M, s);
    launch_fake_table(e->p, e->pools[tgt], e->nobj_total, e->sp.M, s);
    GX_HIP(hipEventRecord(e->pool_ready[tgt], s));
    GX_HIP(hipGetLastError());
    e->pf_valid = true; // the next gx_reset takes it if its key is this one, otherwise it samples inline
    e->pf_key[0] = j.k0; e->pf_key[1] = j.k1;
    return GX_OK;
}

// A stream of the engine's device (ordinary priority: side_stream_priority) for throughput work the caller runs beside
// the stepping -- the tape hand-off's installs and expansions (guardx_amd/dist.py).  ONE per device and process, created on
// first use and never destroyed: the caller's framework may keep per-stream state (torch's caching allocators do, for every
// stream memory was allocated or copied on) that outlives any engine -- a stream that died with an engine crashed the
// interpreter at exit (round 5).  renew: replace the device's stream by a new one (the old one stays alive, so the new one
// gets another hardware queue) -- for a caller that found it sharing a queue with its collective's stream.
static gx_status aux_stream_impl(gx_engine* e, void** stream, bool renew)
{
    if (!e || !stream) return fail(GX_ERR_ARG, "gx_aux_stream: null argument");
    static std::mutex mu;
    static hipStream_t per_device[64] = {};
    if (e->device < 0 || e->device >= 64) return fail(GX_ERR_ARG, "gx_aux_stream: device index out of range");
    std::lock_guard<std::mutex> lock(mu);
    if (!per_device[e->device] || renew) {
        DeviceGuard guard(e->device);
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi); // lo = least urgent
        hipStream_t fresh = nullptr;
        GX_HIP(create_side_stream(&fresh, side_stream_priority("GX_AUX_PRIORITY", 0, lo, hi)));
        per_device[e->device] = fresh;
    }
    *stream = (void*)per_device[e->device];
    e->aux_in_use = true; // from now on the sampler runs at the hand-off's priority (side_stream_priority)
    GX_HIP(ensure_side_ord(e));
    return GX_OK;
}

extern "C" gx_status gx_aux_stream(gx_engine* e, void** stream) { return aux_stream_impl(e, stream, false); }
extern "C" gx_status gx_aux_stream_renew(gx_engine* e, void** stream) { return aux_stream_impl(e, stream, true); }

extern "C" gx_status gx_set_prefetch(gx_engine* e, int32_t steps)
{
    if (!e) return fail(GX_ERR_ARG, "null engine");
    e->prefetch_steps = steps < -2 ? -1 : steps;
    return GX_OK;
}

extern "C" gx_status gx_prefetch_stats(const gx_engine* e, int32_t* hits, int32_t* misses, int32_t* horizon)
{
    if (!e || !hits || !misses || !horizon) return fail(GX_ERR_ARG, "null argument");
    *hits = e->pf_hits; *misses = e->pf_misses;
    *horizon = e->prefetch_steps == -2 ? (e->last_interval > 0 ? e->last_interval : e->cfg.num_steps) : e->prefetch_steps;
    return GX_OK;
}

extern "C" gx_status gx_layout_size(gx_engine* e, int32_t* out)
{
    if (!e || !out) return fail(GX_ERR_ARG, "null argument");
    if (!e->have_reset) return fail(GX_ERR_STATE, "gx_layout_size before gx_reset");
    DeviceGuard guard(e->device);
    if (e->layout_pending) {
        GX_HIP(hipEventSynchronize(e->layout_ev));
        e->layout_pending = false;
    }
    *out = *e->h_layout_size;
    if (*out <= e->cfg.env_total) {
        char buf[128];
        snprintf(buf, sizeof buf, "layout_size %d <= env_num %d (engine.py:444)", *out, e->cfg.env_total);
        return fail(GX_ERR_LAYOUT, buf);
    }
    return GX_OK;
}

extern "C" gx_status gx_layout_size_min(gx_engine* e, int32_t* out)
{
    if (!e || !out) return fail(GX_ERR_ARG, "null argument");
    if (!e->have_reset) return fail(GX_ERR_STATE, "gx_layout_size_min before gx_reset");
    DeviceGuard guard(e->device);
    if (e->layout_pending) {
        GX_HIP(hipEventSynchronize(e->layout_ev));
        e->layout_pending = false;
    }
    *out = e->h_layout_size[1];
    e->h_layout_size[1] = 0x7fffffff;
    if (*out <= e->cfg.env_total) {
        char buf[160];
        snprintf(buf, sizeof buf, "a reset since the last check had layout_size %d <= env_num %d (engine.py:444)", *out,
                 e->cfg.env_total);
        return fail(GX_ERR_LAYOUT, buf);
    }
    return GX_OK;
}

static gx_status step_impl(gx_engine* e, const float* d_action, float* d_obs, float* d_reward, float* d_cost,
                           float* d_done, float* d_qacc, float* d_obs_rd, int32_t* speculated, void* stream)
{
    if (!e || !d_action || !d_obs || !d_reward || !d_cost || !d_done) return fail(GX_ERR_ARG, "null argument");
    if (!e->have_reset) return fail(GX_ERR_STATE, "gx_step before gx_reset (engine.py: _data is None)");
    if ((reinterpret_cast<uintptr_t>(d_action) & 7u) || (reinterpret_cast<uintptr_t>(d_obs) & 3u))
        return fail(GX_ERR_ARG, "d_action must be 8-byte aligned (float2 rows), d_obs 4-byte");
    DeviceGuard guard(e->device);
    hipStream_t s = (hipStream_t)stream;
    // update_data: key, _ = split(key, 2)  engine.py:431
    uint32_t a0, a1, b0, b1;
    split2(e->key[0], e->key[1], a0, a1, b0, b1);
    e->key[0] = a0;
    e->key[1] = a1;
    e->steps_since_reset++;
    e->p.have_last = e->hist >= 1;
    e->p.have_last_last = e->hist >= 2;
    e->last_policy = false;
    if (speculated) *speculated = 0;
    if (use_group_path(e)) {
        RolloutArgs r;
        memset(&r, 0, sizeof r);
        r.T = 1; r.do_reset = 0; r.nobj_total = e->nobj_total; r.hist0 = e->hist;
        r.commit = take_commit(e);
        r.obs_stride = e->p.D; r.sc_stride = 1;
        r.act = static_cast<const float*>(d_action);
        r.obs = d_obs; r.rew = d_reward; r.cost = d_cost; r.done = d_done; r.qacc = d_qacc;
        r.rd_j = e->b.rd_j;
        r.layout_size = e->b.pool.layout_size; r.cand_of = e->b.pool.cand_of; r.cand_xy = e->b.pool.cand_xy;
        r.n_rows = e->sp.M; r.stamps = e->stamps; r.fake = e->b.pool.fake;
        if (d_obs_rd) { // also what reset_done() would return and install, with the key it would use (:447,500)
            uint32_t k[4];
            layout_keys(e, k);
            r.do_reset = 2; r.obs_rd = d_obs_rd; r.keys = nullptr; r.key0 = make_uint4(k[0], k[1], k[2], k[3]);
            e->spec_valid = true;
            if (speculated) *speculated = 1;
        }
        launch_group_rollout(e->p, r, e->b, s);
    } else {
        gx_status st = flush_pending(e, s);
        if (st != GX_OK) return st;
        launch_step(e->p, e->b, d_action, d_obs, d_reward, d_cost, d_done, d_qacc, s);
    }
    if (e->hist < 2) e->hist++;
    GX_HIP(hipGetLastError());
    return GX_OK;
}

extern "C" gx_status gx_step(gx_engine* e, const float* d_action, float* d_obs, float* d_reward,
                             float* d_cost, float* d_done, float* d_qacc, void* stream)
{
    return step_impl(e, d_action, d_obs, d_reward, d_cost, d_done, d_qacc, nullptr, nullptr, stream);
}

extern "C" gx_status gx_step_rd(gx_engine* e, const float* d_action, float* d_obs, float* d_reward,
                                float* d_cost, float* d_done, float* d_qacc, float* d_obs_rd,
                                int32_t* speculated, void* stream)
{
    if (!d_obs_rd || !speculated) return fail(GX_ERR_ARG, "gx_step_rd: null d_obs_rd / speculated");
    return step_impl(e, d_action, d_obs, d_reward, d_cost, d_done, d_qacc, d_obs_rd, speculated, stream);
}

// ---- step() outputs addressed inside a caller-owned slab: one pointer + a slot index per call -------------------------
// An unmodified learner drives Engine.step() once per control step (trpo.py:479-547) and must be handed tensors nobody
// overwrites later (engine.py:495).  The Python host carves them out of one allocation per ~100 calls; passing SIX
// addresses per call through ctypes is a third of that call's host time at env_num = 2000.  Layout of one output set
// (floats; Dp = D rounded up to 4, Np = env_num rounded up to 4, every piece 16-byte aligned):
//   obs [N][D] (at 0) | obs_rd [N][D] (at N*Dp) | reward [N] (at 2*N*Dp) | cost [N] (+Np) | done [N] (+2*Np) | qacc [N][nv] (+3*Np)
extern "C" gx_status gx_step_set_floats(const gx_engine* e, int64_t* floats)
{
    if (!e || !floats) return fail(GX_ERR_ARG, "null argument");
    const int64_t N = e->p.N, Dp = (e->p.D + 3) / 4 * 4, Np = (N + 3) / 4 * 4;
    *floats = 2 * N * Dp + 3 * Np + Np * e->nv;
    return GX_OK;
}

extern "C" gx_status gx_step_slab(gx_engine* e, const float* d_action, float* d_slab, int32_t slot, int32_t flags,
                                  int32_t* speculated, void* stream)
{
    if (!e || !d_slab || slot < 0 || !speculated) return fail(GX_ERR_ARG, "gx_step_slab: bad argument");
    if (reinterpret_cast<uintptr_t>(d_slab) & 15u) return fail(GX_ERR_ARG, "gx_step_slab: d_slab must be 16-byte aligned");
    const int64_t N = e->p.N, Dp = (e->p.D + 3) / 4 * 4, Np = (N + 3) / 4 * 4;
    float* b = d_slab + (size_t)slot * (size_t)(2 * N * Dp + 3 * Np + Np * e->nv);
    float* rew = b + 2 * N * Dp;
    return step_impl(e, d_action, b, rew, rew + Np, rew + 2 * Np, (flags & 1) ? rew + 3 * Np : nullptr,
                     (flags & 2) ? b + N * Dp : nullptr, speculated, stream);
}

extern "C" gx_status gx_reset_done_commit(gx_engine* e)
{
    if (!e) return fail(GX_ERR_ARG, "null engine");
    if (!e->spec_valid && !e->pending_commit)
        return fail(GX_ERR_STATE, "gx_reset_done_commit: the last hot-path call was not a speculating gx_step_rd");
    e->pending_commit = true; // installed by the next launch; idempotent like Engine.reset_done
    return GX_OK;
}

extern "C" gx_status gx_reset_done(gx_engine* e, const float* d_obs_in, float* d_obs_out, void* stream)
{
    if (!e || !d_obs_in || !d_obs_out) return fail(GX_ERR_ARG, "null argument");
    if (!e->have_reset) return fail(GX_ERR_STATE, "gx_reset_done before gx_reset");
    DeviceGuard guard(e->device);
    hipStream_t s = (hipStream_t)stream;
    if (e->hist == 0) { // self._done is None: mjx_reset_done falls through (engine.py:713)
        if (d_obs_in != d_obs_out)
            GX_HIP(hipMemcpyAsync(d_obs_out, d_obs_in, sizeof(float) * (size_t)e->p.N * e->p.D,
                                  hipMemcpyDeviceToDevice, s));
        return GX_OK;
    }
    uint32_t k[4];
    layout_keys(e, k);
    { gx_status st = flush_pending(e, s); if (st != GX_OK) return st; }
    launch_reset_done(e->p, e->b, e->nobj_total, k[0], k[1], k[2], k[3], d_obs_in, d_obs_out, s);
    GX_HIP(hipGetLastError());
    return GX_OK;
}

// Per-step layout keys of a T-step fused rollout, staged in pinned device-visible host memory: step t advances the key
// (engine.py:431), the reset_done that follows draws randint with that key (engine.py:447,500).  Returns the slot.
//
// When may a slot be written again?  Rounds 1-4 kept four slots and an event per slot, recorded behind the launch that read
// it, and had the host wait for the event of the slot it was about to reuse -- four launches old, so "never a wait".  On
// this ROCm (7.2) that wait is one: hipEventQuery answers hipErrorNotReady for EVERY such event, however old (timing
// enabled or not), and hipEventSynchronize then returns when everything queued on the stream so far has run -- measured in
// round 5 (tools/ab/profile_rehearsal_host.py, -DGX_HOST_TIMING): 1.1-1.4 ms per call inside an Ant epoch, the host a whole
// dynamics pass behind the GPU in every epoch; with one stream that costs a launch latency per epoch, with the tape
// hand-off's several streams (the N > 1 configuration) it left the GPU idle for 140-340 us per epoch.  So: no events.  The
// ring is long (up to 1024 slots, 8 MB), and the host synchronises the streams it handed slots to ONCE PER LAP.
static gx_status stage_rollout_keys(gx_engine* e, int32_t T, hipStream_t s, int& slot_out, uint32_t& k0_out, uint32_t& k1_out)
{
    auto drain = [&]() -> hipError_t {
        hipError_t err = hipSuccess;
        for (hipStream_t q : e->keys_streams)
            if (err == hipSuccess) err = hipStreamSynchronize(q);
        e->keys_streams.clear();
        if (err != hipSuccess) { // a stream the caller has destroyed since (its work is done or gone with it): the device, then
            (void)hipGetLastError();
            err = hipDeviceSynchronize();
        }
        return err;
    };
    if (e->keys_cap < T) {
        GX_HIP(drain()); // launches still reading the old ring
        if (e->h_keys) (void)hipHostFree(e->h_keys);
        e->h_keys = nullptr; e->keys_cap = 0; e->keys_slots = 0; e->keys_next = 0;
        const int cap = T > 256 ? T : 256;
        size_t slots = ((size_t)8 << 20) / (sizeof(uint4) * (size_t)cap);
        const char* fv = getenv("GX_KEY_RING"); // experiments / tests (read at every allocation: they are rare)
        const int forced = fv ? atoi(fv) : 0;
        slots = forced >= 2 ? (size_t)forced : (slots < 16 ? 16 : (slots > 1024 ? 1024 : slots));
        GX_HIP(hipHostMalloc((void**)&e->h_keys, sizeof(uint4) * (size_t)cap * slots, hipHostMallocMapped));
        e->keys_cap = cap; e->keys_slots = (int)slots;
    }
    const int slot = e->keys_next;
    if (slot == 0) GX_HIP(drain()); // a lap is over: every launch that read the ring has to be
    e->keys_next = (slot + 1) % e->keys_slots;
    bool known = false;
    for (hipStream_t q : e->keys_streams) known = known || q == s;
    if (!known) e->keys_streams.push_back(s);
    uint4* keys = e->h_keys + (size_t)slot * e->keys_cap;
    uint32_t k0 = e->key[0], k1 = e->key[1];
    for (int32_t t = 0; t < T; ++t) {
        uint32_t a0, a1, b0, b1;
        split2(k0, k1, a0, a1, b0, b1);
        k0 = a0; k1 = a1;
        uint4 kk;
        split2(k0, k1, kk.x, kk.y, kk.z, kk.w);
        keys[t] = kk;
    }
    slot_out = slot; k0_out = k0; k1_out = k1;
    return GX_OK;
}

static void fill_rollout_args(gx_engine* e, RolloutArgs& r, int32_t T, int slot)
{
    memset(&r, 0, sizeof r);
    r.T = T; r.do_reset = 1; r.nobj_total = e->nobj_total; r.hist0 = e->hist;
    r.obs_stride = e->p.D; r.sc_stride = 1; r.rd_j = e->b.rd_j;
    r.keys = e->h_keys + (size_t)slot * e->keys_cap; // pinned + device-visible: read over the host link only on a reset
    r.layout_size = e->b.pool.layout_size; r.cand_of = e->b.pool.cand_of; r.cand_xy = e->b.pool.cand_xy;
    r.n_rows = e->sp.M; r.stamps = e->stamps; r.fake = e->b.pool.fake;
    e->p.have_last = e->hist >= 1;
    e->p.have_last_last = e->hist >= 2;
}

static gx_status rollout_impl(gx_engine* e, int32_t T, const float* d_actions, float* d_obs, float* d_reward,
                              float* d_cost, float* d_done, float* d_act_out, int obs_stride, int sc_stride,
                              void* stream)
{
    if (!e->have_reset) return fail(GX_ERR_STATE, "gx_rollout before gx_reset");
    if (reinterpret_cast<uintptr_t>(d_actions) & 7u) return fail(GX_ERR_ARG, "d_actions must be 8-byte aligned");
    DeviceGuard guard(e->device);
    hipStream_t s = (hipStream_t)stream;
    int slot; uint32_t k0, k1;
    gx_status st = stage_rollout_keys(e, T, s, slot, k0, k1);
    if (st != GX_OK) return st;
    RolloutArgs r;
    fill_rollout_args(e, r, T, slot);
    r.act = static_cast<const float*>(d_actions);
    r.obs = d_obs; r.rew = d_reward; r.cost = d_cost; r.done = d_done; r.qacc = nullptr;
    r.act_out = d_act_out; r.obs_stride = obs_stride; r.sc_stride = sc_stride;
    e->last_policy = false;
    if (use_split_rollout(e, T)) { // light robots, small env_num: dynamics tape + one thread per (step, env) row
        const size_t nt = tape_floats_padded(e, T);
        const size_t need = nt + (size_t)e->p.N * split_entry_width(e->p); // [tape | entry records]
        if (need > e->tape_cap) {
            GX_HIP(hipStreamSynchronize(s));            // an earlier launch may still read the old tape
            if (e->tape) (void)hipFree(e->tape);
            e->tape = nullptr; e->tape_cap = 0;
            GX_HIP(hipMalloc((void**)&e->tape, sizeof(float) * need));
            e->tape_cap = need;
        }
        if (!e->obj0) GX_HIP(hipMalloc((void**)&e->obj0, sizeof(float4) * (size_t)e->p.P * e->p.Npad));
        st = flush_pending(e, s);
        if (st != GX_OK) return st;
        hipEvent_t hold = nullptr;
        if (e->pf_phase1_pending && getenv("GX_NO_OBS_HOLD") == nullptr) { hold = e->pf_phase1; e->pf_phase1_pending = false; }
        // a prefetch sampler is in flight beside this rollout: the one-lane dynamics pass (see SwimmerRobot::kDynLanes)
        const int lanes = dyn_pass_has_company(e) ? 1 : 4;
        GX_HIP(launch_split_rollout(e->p, r, e->tape, e->obj0, e->tape + nt, e->b, s, hold, 3, lanes));
    } else if (use_group_path(e)) {   // latency regime: 16 lanes per env
        r.commit = take_commit(e);
        launch_group_rollout(e->p, r, e->b, s);
    } else {                    // bandwidth regime: one thread per env
        st = flush_pending(e, s);
        if (st != GX_OK) return st;
        launch_thread_rollout(e->p, r, e->b, s);
    }
    GX_HIP(hipGetLastError());
    e->key[0] = k0; e->key[1] = k1;
    e->steps_since_reset += T;
    e->hist = (e->hist + T) >= 2 ? 2 : e->hist + T;
    return GX_OK;
}

extern "C" gx_status gx_rollout(gx_engine* e, int32_t T, const float* d_actions, float* d_obs,
                                float* d_reward, float* d_cost, float* d_done, void* stream)
{
    if (!e || !d_actions || !d_obs || !d_reward || !d_cost || !d_done || T < 1)
        return fail(GX_ERR_ARG, "bad argument");
    return rollout_impl(e, T, d_actions, d_obs, d_reward, d_cost, d_done, nullptr, e->p.D, 1, stream);
}

extern "C" gx_status gx_rollout_packed(gx_engine* e, int32_t T, const float* d_actions, float* d_packed,
                                       void* stream)
{
    if (!e || !d_actions || !d_packed || T < 1) return fail(GX_ERR_ARG, "bad argument");
    const int W = e->p.D + e->na + 3;
    return rollout_impl(e, T, d_actions, d_packed, d_packed + e->p.D + e->na, d_packed + e->p.D + e->na + 1,
                        d_packed + e->p.D + e->na + 2, d_packed + e->p.D, W, W, stream);
}

extern "C" int32_t gx_packed_width(const gx_engine* e) { return e ? e->p.D + e->na + 3 : -1; }

// ---------------------------------------------------------------------------------------------------------------
// tape hand-off: the rank that steps the envs runs only the serial dynamics pass and hands out its tape (40 B per
// env-step for the Point -- qpos, qvel, action, done, two layout-row indices -- instead of the 192 B packed row);
// whoever needs the rollout -- every rank, after ONE all-gather of the tapes -- runs the observation pass on it, which
// re-derives pose, ctrl and reward.  Every rank samples the same layout pools (the key is shared, engine.py:263), so
// the pool rows a tape's reset_done events refer to are local everywhere.
// Buffer of one shard: [ tape T*N*W | layouts at entry P*Npad*4 | entry records N*12 ] floats.
// ---------------------------------------------------------------------------------------------------------------
extern "C" gx_status gx_tape_floats(const gx_engine* e, int32_t T, int64_t* tape, int64_t* obj0, int64_t* entry)
{
    if (!e || T < 1 || !tape || !obj0 || !entry) return fail(GX_ERR_ARG, "bad argument");
    if (!split_rollout_supported(e->p))
        return fail(GX_ERR_UNSUPPORTED, "tape hand-off: needs a task without observe_vel / observe_acc and one physics step per control step");
    *tape = (int64_t)tape_floats_padded(e, T);
    *obj0 = (int64_t)e->p.P * e->p.Npad * 4;
    *entry = (int64_t)e->p.N * split_entry_width(e->p);
    return GX_OK;
}

extern "C" gx_status gx_rollout_tape(gx_engine* e, int32_t T, const float* d_actions, float* d_shard,
                                     int64_t* token, void* stream)
{
    if (!e || !d_actions || !d_shard || !token || T < 1) return fail(GX_ERR_ARG, "bad argument");
    if (!e->have_reset) return fail(GX_ERR_STATE, "gx_rollout_tape before gx_reset");
    if (!split_rollout_supported(e->p))
        return fail(GX_ERR_UNSUPPORTED, "tape hand-off: needs a task without observe_vel / observe_acc and one physics step per control step");
    if ((reinterpret_cast<uintptr_t>(d_actions) & 7u) || (reinterpret_cast<uintptr_t>(d_shard) & 15u))
        return fail(GX_ERR_ARG, "d_actions must be 8-byte, d_shard 16-byte aligned");
    DeviceGuard guard(e->device);
    hipStream_t s = (hipStream_t)stream;
#ifdef GX_HOST_TIMING
    auto t_0 = std::chrono::steady_clock::now();
#endif
    gx_status st = flush_pending(e, s); // (before a key slot is taken: nothing to undo)
    if (st != GX_OK) return st;
    int slot; uint32_t k0, k1;
#ifdef GX_HOST_TIMING
    auto t_1 = std::chrono::steady_clock::now();
#endif
    st = stage_rollout_keys(e, T, s, slot, k0, k1);
    if (st != GX_OK) return st;
#ifdef GX_HOST_TIMING
    auto t_2 = std::chrono::steady_clock::now();
#endif
    RolloutArgs r;
    fill_rollout_args(e, r, T, slot);
    r.act = d_actions;
    e->last_policy = false;
    const size_t nt = tape_floats_padded(e, T), no = (size_t)e->p.P * e->p.Npad * 4;
    GX_HIP(launch_split_rollout(e->p, r, d_shard, reinterpret_cast<float4*>(d_shard + nt), d_shard + nt + no, e->b, s,
                                nullptr, 1, dyn_pass_has_company(e) ? 1 : 4));
#ifdef GX_HOST_TIMING // (a variant build: GX_EXTRA_FLAGS_gx_api="-DGX_HOST_TIMING" python tools/build_variant.py hosttiming)
    {
        auto t_3 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        fprintf(stderr, "gx_rollout_tape host us: flush %ld keys %ld launch %ld\n", us(t_0, t_1), us(t_1, t_2), us(t_2, t_3));
    }
#endif
    GX_HIP(hipGetLastError());
    e->key[0] = k0; e->key[1] = k1;
    e->steps_since_reset += T;
    e->hist = (e->hist + T) >= 2 ? 2 : e->hist + T;
    *token = ((int64_t)e->pool_gen[e->cur] << 8) | (int64_t)e->cur;
    return GX_OK;
}

static gx_status expand_impl(gx_engine* e, int32_t T, const float* d_shards, int64_t stride_floats, int32_t n_shards,
                             int64_t token, float* d_packed, int64_t packed_stride_floats, void* stream)
{
    if (!e || !d_shards || !d_packed || T < 1 || n_shards < 1 || n_shards > 65535) return fail(GX_ERR_ARG, "bad argument");
    if (!split_rollout_supported(e->p))
        return fail(GX_ERR_UNSUPPORTED, "tape hand-off: needs a task without observe_vel / observe_acc and one physics step per control step");
    if ((reinterpret_cast<uintptr_t>(d_shards) & 15u) || (n_shards > 1 && (stride_floats & 3)))
        return fail(GX_ERR_ARG, "d_shard must be 16-byte aligned (and the shards a multiple of 4 floats apart)");
    const int pi = (int)(token & 0xff);
    if (pi < 0 || pi >= gx_engine::kPools || (uint32_t)(token >> 8) != e->pool_gen[pi])
        return fail(GX_ERR_STATE, "gx_expand_tape: the layout pool of this tape has been resampled (expand a tape "
                                  "before the second gx_reset after its rollout)");
    DeviceGuard guard(e->device);
    hipStream_t s = (hipStream_t)stream;
    RolloutArgs r;
    memset(&r, 0, sizeof r);
    const int W = e->p.D + e->na + 3;
    r.T = T; r.do_reset = 1; r.nobj_total = e->nobj_total;
    r.cand_xy = e->pools[pi].cand_xy; r.n_rows = e->sp.M; r.fake = e->pools[pi].fake;
    const size_t nt = tape_floats_padded(e, T), no = (size_t)e->p.P * e->p.Npad * 4;
    if (n_shards > 1 && ((size_t)stride_floats < nt + no + (size_t)e->p.N * split_entry_width(e->p) ||
                         (size_t)packed_stride_floats < (size_t)T * e->p.N * W))
        return fail(GX_ERR_ARG, "gx_expand_tapes: the strides are smaller than one shard / one packed rollout");
    r.act = nullptr; // the tape rows carry the actions
    r.obs = d_packed; r.act_out = d_packed + e->p.D;
    r.rew = d_packed + e->p.D + e->na; r.cost = r.rew + 1; r.done = r.rew + 2;
    r.obs_stride = W; r.sc_stride = W;
    // the pool must be complete on this stream (it is when the tape's rank has stepped, but this may be another stream)
    GX_HIP(hipStreamWaitEvent(s, e->pool_ready[pi], 0));
    GX_HIP(launch_split_rollout(e->p, r, const_cast<float*>(d_shards),
                                reinterpret_cast<float4*>(const_cast<float*>(d_shards) + nt),
                                const_cast<float*>(d_shards) + nt + no, e->b, s, nullptr, 2, 1, n_shards, stride_floats,
                                packed_stride_floats));
    GX_HIP(hipEventRecord(e->expand_ev[pi], s));
    e->expand_pending[pi] = true;
    GX_HIP(hipGetLastError());
    return GX_OK;
}

extern "C" gx_status gx_expand_tape(gx_engine* e, int32_t T, const float* d_shard, int64_t token, float* d_packed,
                                    void* stream)
{
    return expand_impl(e, T, d_shard, 0, 1, token, d_packed, 0, stream);
}

// the observation pass over the shards of ALL ranks in one launch (the all-gathered buffer as it is)
extern "C" gx_status gx_expand_tapes(gx_engine* e, int32_t T, const float* d_shards, int64_t stride_floats, int32_t n_shards,
                                     int64_t token, float* d_packed, int64_t packed_stride_floats, void* stream)
{
    return expand_impl(e, T, d_shards, stride_floats, n_shards, token, d_packed, packed_stride_floats, stream);
}

// Hidden widths whose weights do not fit the fused kernel's LDS (128, 256; also 64 with gx_set_policy_impl(e, 3), as a
// cross-check): per control step one policy launch over all envs (gx_policy_step.hip) and the ordinary fused
// step + reset_done launch -- the loop of oracle/gx_oracle.c:gxo_rollout_policy, the arithmetic of the fused kernel.
static gx_status rollout_policy_stepwise(gx_engine* e, int32_t T, const gx_policy* pol, const float* d_obs0, float* d_obs_in,
                                         float* d_act, float* d_logp, float* d_val, float* d_mu, float* d_reward,
                                         float* d_cost, float* d_done, float* d_obs_last, float* d_val_last, float* d_logstd,
                                         void* stream)
{
    const int H = pol->hidden;
    if (!policy_step_supported(H))
        return fail(GX_ERR_UNSUPPORTED, "gx_rollout_policy: hidden_sizes must be (64, 64), (128, 128), (192, 192) or (256, 256)");
    if ((e->na & 1) || e->na > 16) return fail(GX_ERR_UNSUPPORTED, "gx_rollout_policy: needs an even action width <= 16");
    if (reinterpret_cast<uintptr_t>(d_act) & 7u) return fail(GX_ERR_ARG, "gx_rollout_policy: d_act must be 8-byte aligned");
    DeviceGuard guard(e->device);
    hipStream_t s = (hipStream_t)stream;
    const int N = e->p.N, D = e->p.D, A = e->na;
    const size_t need = (size_t)policy_step_wt_floats(D, H);
    if (need > e->pol_wt_cap) {
        GX_HIP(hipStreamSynchronize(s));
        if (e->pol_wt) (void)hipFree(e->pol_wt);
        e->pol_wt = nullptr; e->pol_wt_cap = 0;
        GX_HIP(hipMalloc((void**)&e->pol_wt, sizeof(float) * need));
        e->pol_wt_cap = need;
    }
    if (!e->pol_cur) GX_HIP(hipMalloc((void**)&e->pol_cur, sizeof(float) * (size_t)N * D));
    const bool group = use_group_path(e);
    const bool valu = e->policy_impl == 1; // gx_set_policy_impl(e, 1): fmaf chains; otherwise the MFMA tiles
    gx_status st = GX_OK;
    if (!group) { st = flush_pending(e, s); if (st != GX_OK) return st; } // (before a key slot is taken: nothing to undo)
    int slot; uint32_t k0, k1;
    st = stage_rollout_keys(e, T, s, slot, k0, k1);
    if (st != GX_OK) return st;
    launch_policy_transpose(pol->d_params, e->pol_wt, D, A, H, s);
    GX_HIP(hipMemcpyAsync(e->pol_cur, d_obs0, sizeof(float) * (size_t)N * D, hipMemcpyDeviceToDevice, s));
    for (int32_t t = 0; t < T; ++t) {
        const size_t tn = (size_t)t * N;
        launch_policy_step(H, pol->d_params, e->pol_wt, e->pol_cur, pol->seed[0], pol->seed[1], e->policy_steps + (uint32_t)t, N, D,
                           A, e->p.env_offset, 0, d_obs_in + tn * D, d_act + tn * A, d_mu + tn * A, d_logp + tn, d_val + tn,
                           nullptr, d_logstd, s, valu);
        RolloutArgs r;
        fill_rollout_args(e, r, 1, slot);
        r.keys = e->h_keys + (size_t)slot * e->keys_cap + t; // this step's reset_done key (engine.py:431,447,500)
        r.act = d_act + tn * A;
        r.obs = e->pol_cur;                    // the post-reset_done observation feeds the next policy step (trpo.py:547)
        r.rew = d_reward + tn; r.cost = d_cost + tn; r.done = d_done + tn; r.qacc = nullptr;
        if (group) {
            r.commit = t == 0 ? take_commit(e) : 0;
            launch_group_rollout(e->p, r, e->b, s);
        } else {
            launch_thread_rollout(e->p, r, e->b, s);
        }
        if (e->hist < 2) e->hist++;
    }
    launch_policy_step(H, pol->d_params, e->pol_wt, e->pol_cur, pol->seed[0], pol->seed[1], 0u, N, D, A, e->p.env_offset, 1,
                       nullptr, nullptr, nullptr, nullptr, d_val_last, d_obs_last, nullptr, s, valu);
    e->last_policy = false; // (the open-loop kernels ran: the prefetch sampler keeps its back-to-back chain)
    GX_HIP(hipGetLastError());
    e->key[0] = k0; e->key[1] = k1;
    e->steps_since_reset += T;
    e->policy_steps += (uint32_t)T;
    return GX_OK;
}

extern "C" gx_status gx_rollout_policy(gx_engine* e, int32_t T, const gx_policy* pol, const float* d_obs0,
                                       float* d_obs_in, float* d_act, float* d_logp, float* d_val,
                                       float* d_mu, float* d_reward, float* d_cost, float* d_done,
                                       float* d_obs_last, float* d_val_last, float* d_logstd, void* stream)
{
    if (!e || !pol || T < 1 || !d_obs0 || !d_obs_in || !d_act || !d_logp || !d_val || !d_mu || !d_reward ||
        !d_cost || !d_done || !d_obs_last || !d_val_last || !d_logstd)
        return fail(GX_ERR_ARG, "gx_rollout_policy: bad argument");
    if (pol->struct_size != (int32_t)sizeof(gx_policy) || !pol->d_params)
        return fail(GX_ERR_ARG, "gx_policy.struct_size mismatch or null parameters");
    if (!e->have_reset) return fail(GX_ERR_STATE, "gx_rollout_policy before gx_reset");
    // hidden_sizes (128, 128) in ONE launch (round 5: hidden-layer weights resident in registers, gx_policy.h) -- the
    // default for the light robots' default-width observations; gx_set_policy_impl(e, 1 | 2 | 3) keeps the step-wise forms
    // ... and (192, 192), (256, 256) with the hidden-layer weights streamed from their L2-resident transposed copy
    const bool fusedw = (pol->hidden == kPolHd2 || pol->hidden == 192 || pol->hidden == 256) && e->policy_impl == 0 &&
                        policy_fused128_supported(e->p) && !(e->na & 1) && e->p.N <= 65536;
    const bool fused128 = fusedw && pol->hidden == kPolHd2;
    if ((pol->hidden != kPolHd && !fusedw) || e->policy_impl == 3)
        return rollout_policy_stepwise(e, T, pol, d_obs0, d_obs_in, d_act, d_logp, d_val, d_mu, d_reward, d_cost, d_done,
                                       d_obs_last, d_val_last, d_logstd, stream);
    if (!policy_rollout_supported(e->p) || (e->na & 1) || e->na > 16 || e->p.N > 65536)
        return fail(GX_ERR_UNSUPPORTED, "gx_rollout_policy: needs hazards_num <= 15, lidar_num_bins <= 16, env_num <= 65536");
    const int impl = fused128 ? 3 : (fusedw ? pol->hidden : (e->policy_impl == 1 ? 1 : 2)); // auto = MFMA
    if (policy_lds_bytes(e->p, impl) > 150 * 1024)
        return fail(GX_ERR_UNSUPPORTED, "gx_rollout_policy: observation too wide for the LDS-resident weights");
    DeviceGuard guard(e->device);
    hipStream_t s = (hipStream_t)stream;
    int slot; uint32_t k0, k1;
    gx_status st = stage_rollout_keys(e, T, s, slot, k0, k1);
    if (st != GX_OK) return st;
    RolloutArgs r;
    fill_rollout_args(e, r, T, slot);
    r.rew = d_reward; r.cost = d_cost; r.done = d_done;
    PolicyArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.params = pol->d_params; pa.seed0 = pol->seed[0]; pa.seed1 = pol->seed[1]; pa.t0 = e->policy_steps;
    pa.obs0 = d_obs0; pa.obs_in = d_obs_in; pa.act = d_act; pa.logp = d_logp; pa.val = d_val; pa.mu = d_mu;
    pa.obs_last = d_obs_last; pa.val_last = d_val_last; pa.logstd = d_logstd;
    if (fusedw && !fused128) { // the streaming form reads the [k][unit] transposed hidden layers (as the step-wise form does)
        const size_t need = (size_t)policy_step_wt_floats(e->p.D, pol->hidden);
        if (need > e->pol_wt_cap) {
            GX_HIP(hipStreamSynchronize(s));
            if (e->pol_wt) (void)hipFree(e->pol_wt);
            e->pol_wt = nullptr; e->pol_wt_cap = 0;
            GX_HIP(hipMalloc((void**)&e->pol_wt, sizeof(float) * need));
            e->pol_wt_cap = need;
        }
        launch_policy_transpose(pol->d_params, e->pol_wt, e->p.D, e->na, pol->hidden, s);
        pa.wt = e->pol_wt;
    }
    r.commit = take_commit(e);
    launch_policy_rollout(e->p, r, pa, e->b, impl, s);
    e->last_policy = true;
    GX_HIP(hipGetLastError());
    e->key[0] = k0; e->key[1] = k1;
    e->steps_since_reset += T;
    e->hist = (e->hist + T) >= 2 ? 2 : e->hist + T;
    e->policy_steps += (uint32_t)T;
    return GX_OK;
}

extern "C" gx_status gx_set_policy_impl(gx_engine* e, int32_t impl)
{
    if (!e || impl < 0 || impl > 3) return fail(GX_ERR_ARG, "gx_set_policy_impl: 0 auto, 1 VALU, 2 MFMA, 3 step-wise");
    e->policy_impl = impl;
    return GX_OK;
}

extern "C" gx_status gx_math_probe2(int32_t n, const float* d_x, float* d_log, float* d_tanh, void* stream)
{
    if (n < 1 || !d_x || !d_log || !d_tanh) return fail(GX_ERR_ARG, "bad argument");
    launch_math_probe2(n, d_x, d_log, d_tanh, (hipStream_t)stream);
    GX_HIP(hipGetLastError());
    return GX_OK;
}

// ---------------------------------------------------------------------------
// state exchange (tests / checkpoints); synchronous.  The dynamic state of env i is the
// flat sequence  q[nq] v[nv] pose0[4] done0 steps  spread over ndyn float4 arrays.
// ---------------------------------------------------------------------------
static inline float& dyn_at(std::vector<float4>& dyn, size_t Np, int i, int k)
{
    float4& f = dyn[(size_t)(k >> 2) * Np + i];
    return (k & 3) == 0 ? f.x : ((k & 3) == 1 ? f.y : ((k & 3) == 2 ? f.z : f.w));
}

extern "C" gx_status gx_get_state(gx_engine* e, float* qpos, float* qvel, float* pose0, float* pose1,
                                  float* objs, float* done0, float* done1, float* steps,
                                  uint32_t* key, int32_t* hist)
{
    if (!e) return fail(GX_ERR_ARG, "null engine");
    DeviceGuard guard(e->device);
    GX_HIP(hipDeviceSynchronize());
    if (e->pending_commit) { // a requested reset_done is part of the state
        gx_status st = flush_pending(e, nullptr);
        if (st != GX_OK) return st;
        GX_HIP(hipDeviceSynchronize());
    }
    const Params& p = e->p;
    const size_t Np = p.Npad;
    const int nq = e->nq, nv = e->nv;
    std::vector<float4> dyn((size_t)e->ndyn * Np), obj((size_t)p.P * Np), hs(Np);
    GX_HIP(hipMemcpy(dyn.data(), e->b.dyn, sizeof(float4) * dyn.size(), hipMemcpyDeviceToHost));
    GX_HIP(hipMemcpy(obj.data(), e->b.obj, sizeof(float4) * obj.size(), hipMemcpyDeviceToHost));
    GX_HIP(hipMemcpy(hs.data(), e->b.hist, sizeof(float4) * hs.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < p.N; ++i) {
        if (qpos) for (int k = 0; k < nq; ++k) qpos[(size_t)i * nq + k] = dyn_at(dyn, Np, i, k);
        if (qvel) for (int k = 0; k < nv; ++k) qvel[(size_t)i * nv + k] = dyn_at(dyn, Np, i, nq + k);
        if (pose0) for (int k = 0; k < 4; ++k) pose0[4 * i + k] = dyn_at(dyn, Np, i, nq + nv + k);
        if (pose1) { pose1[2 * i] = hs[i].x; pose1[2 * i + 1] = hs[i].y; }
        if (done0) done0[i] = dyn_at(dyn, Np, i, nq + nv + 4);
        if (done1) done1[i] = hs[i].z;
        if (steps) steps[i] = dyn_at(dyn, Np, i, nq + nv + 5);
        if (objs)
            for (int o = 0; o < p.nobj; ++o) {
                const float4 v = obj[(size_t)(o / 2) * Np + i];
                objs[((size_t)i * p.nobj + o) * 2] = (o & 1) ? v.z : v.x;
                objs[((size_t)i * p.nobj + o) * 2 + 1] = (o & 1) ? v.w : v.y;
            }
    }
    if (key) { key[0] = e->key[0]; key[1] = e->key[1]; }
    if (hist) *hist = e->hist;
    return GX_OK;
}

extern "C" gx_status gx_set_state(gx_engine* e, const float* qpos, const float* qvel, const float* pose0,
                                  const float* pose1, const float* objs, const float* done0,
                                  const float* done1, const float* steps, const uint32_t* key,
                                  const int32_t* hist)
{
    if (!e) return fail(GX_ERR_ARG, "null engine");
    DeviceGuard guard(e->device);
    GX_HIP(hipDeviceSynchronize());
    if (e->pending_commit) { // a requested reset_done is part of the state
        gx_status st = flush_pending(e, nullptr);
        if (st != GX_OK) return st;
        GX_HIP(hipDeviceSynchronize());
    }
    const Params& p = e->p;
    const size_t Np = p.Npad;
    const int nq = e->nq, nv = e->nv;
    std::vector<float4> dyn((size_t)e->ndyn * Np), obj((size_t)p.P * Np), hs(Np);
    GX_HIP(hipMemcpy(dyn.data(), e->b.dyn, sizeof(float4) * dyn.size(), hipMemcpyDeviceToHost));
    GX_HIP(hipMemcpy(obj.data(), e->b.obj, sizeof(float4) * obj.size(), hipMemcpyDeviceToHost));
    GX_HIP(hipMemcpy(hs.data(), e->b.hist, sizeof(float4) * hs.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < p.N; ++i) {
        if (qpos) for (int k = 0; k < nq; ++k) dyn_at(dyn, Np, i, k) = qpos[(size_t)i * nq + k];
        if (qvel) for (int k = 0; k < nv; ++k) dyn_at(dyn, Np, i, nq + k) = qvel[(size_t)i * nv + k];
        if (pose0) for (int k = 0; k < 4; ++k) dyn_at(dyn, Np, i, nq + nv + k) = pose0[4 * i + k];
        if (pose1) { hs[i].x = pose1[2 * i]; hs[i].y = pose1[2 * i + 1]; }
        if (done0) dyn_at(dyn, Np, i, nq + nv + 4) = done0[i];
        if (done1) hs[i].z = done1[i];
        if (steps) dyn_at(dyn, Np, i, nq + nv + 5) = steps[i];
        if (objs)
            for (int o = 0; o < p.nobj; ++o) {
                float4& v = obj[(size_t)(o / 2) * Np + i];
                const float ox = objs[((size_t)i * p.nobj + o) * 2], oy = objs[((size_t)i * p.nobj + o) * 2 + 1];
                if (o & 1) { v.z = ox; v.w = oy; } else { v.x = ox; v.y = oy; }
            }
    }
    GX_HIP(hipMemcpy(e->b.dyn, dyn.data(), sizeof(float4) * dyn.size(), hipMemcpyHostToDevice));
    GX_HIP(hipMemcpy(e->b.obj, obj.data(), sizeof(float4) * obj.size(), hipMemcpyHostToDevice));
    GX_HIP(hipMemcpy(e->b.hist, hs.data(), sizeof(float4) * hs.size(), hipMemcpyHostToDevice));
    if (key) { e->key[0] = key[0]; e->key[1] = key[1]; }
    if (hist) e->hist = *hist;
    e->have_reset = true;
    e->spec_valid = false;
    return GX_OK;
}

extern "C" gx_status gx_get_pool(gx_engine* e, float* pool, int32_t max_rows, int32_t* got)
{
    if (!e || !pool || !got) return fail(GX_ERR_ARG, "null argument");
    if (!e->have_reset) return fail(GX_ERR_STATE, "gx_get_pool before gx_reset");
    DeviceGuard guard(e->device);
    GX_HIP(hipDeviceSynchronize());
    int L = 0;
    GX_HIP(hipMemcpy(&L, e->b.pool.layout_size, sizeof(int), hipMemcpyDeviceToHost));
    const int n = L < max_rows ? L : max_rows;
    std::vector<int> idx(n > 0 ? n : 1);
    if (n > 0) GX_HIP(hipMemcpy(idx.data(), e->b.pool.cand_of, sizeof(int) * n, hipMemcpyDeviceToHost));
    const size_t row = (size_t)e->nobj_total * 2;
    for (int r = 0; r < n; ++r)
        GX_HIP(hipMemcpy(pool + r * row, e->b.pool.cand_xy + (size_t)idx[r] * e->nobj_total, sizeof(float) * row,
                         hipMemcpyDeviceToHost));
    *got = n;
    return GX_OK;
}

extern "C" gx_status gx_debug_stamps(gx_engine* e, uint64_t* d_stamps)
{
    if (!e) return fail(GX_ERR_ARG, "null engine");
    e->stamps = reinterpret_cast<unsigned long long*>(d_stamps);
    return GX_OK;
}

extern "C" gx_status gx_set_path(gx_engine* e, int32_t mode)
{
    if (!e || mode < 0 || mode > 3)
        return fail(GX_ERR_ARG, "gx_set_path: mode must be 0 (auto), 1 (thread-per-env), 2 (lane-group) or 3 (two-kernel "
                                "rollouts where supported, lane-group otherwise)");
    e->path_mode = mode;
    return GX_OK;
}

extern "C" gx_status gx_math_probe(int32_t n, const float* d_x, const float* d_y, float* d_s, float* d_c,
                                   float* d_at2, float* d_ex, void* stream)
{
    if (n < 1 || !d_x || !d_y || !d_s || !d_c || !d_at2 || !d_ex) return fail(GX_ERR_ARG, "bad argument");
    launch_math_probe(n, d_x, d_y, d_s, d_c, d_at2, d_ex, (hipStream_t)stream);
    GX_HIP(hipGetLastError());
    return GX_OK;
}

extern "C" gx_status gx_split_probe(const uint32_t* key, int32_t n, uint32_t* d_out, void* stream)
{
    if (!key || n < 1 || !d_out) return fail(GX_ERR_ARG, "bad argument");
    launch_split_probe(key[0], key[1], n, d_out, (hipStream_t)stream);
    GX_HIP(hipGetLastError());
    return GX_OK;
}
