// gx_kernels.h -- host-callable launchers of the HIP kernels (gx_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gx_device.h"
#include "gx_policy.h"

namespace gx {

// layout-sampler parameters (engine.py:546-621); object types: 0 goal, 1 hazard, 2 robot, 3 pillar (synthetic
// extension); placement order goal, hazards, pillars, robot
struct SampleParams {
    int M;          // candidates sampled by this launch (engine.py:263: all 1e6; a shard: its share of them)
    int c0, Mtot;   // ... which are candidates c0 .. c0 + M - 1 of the Mtot the reference draws (split(key, Mtot)[c])
    int nobj_total; // goal + hazards + pillars + robot
    int H;          // hazards: objects 1..H; pillars: H+1 .. nobj_total-2
    float lo_x[4], hi_x[4], lo_y[4], hi_y[4]; // by type (hazards / pillars: default rectangle)
    const float4* haz_bounds;                  // null, or per hazard-or-pillar (lox, hix, loy, hiy), index o-1
    float thr[4][4]; // thr[placed type][new type] = f32(keepout_p + margin + keepout_new)
    float min_rg;    // engine.py:571
    float thr_sq[4][4]; // exact cutoffs: sqrtf(d2) < thr  <=>  d2 < thr_sq
    float min_rg_sq;
    uint32_t k0, k1; // engine key at reset time
    int fused;       // 1: sparse arena -- phases 0 and 2 only, the robot drawn at the end of the ONE chain walk (gx_kernels.hip)
    unsigned long long* dbg; // null, or per phase-2 wave: s_memtime at entry / exit, HW_ID, XCC_ID (gx_debug_stamps)
};


// valid-layout pool of one reset_layout() (engine.py:433-444); two of them are kept so the
// next epoch's pool can be sampled on a side stream while the current one is in use
struct Pool {
    uint8_t* cand_ok;  // [M padded to whole compaction tiles] 0 / 1 (the padding stays 0)
    float2* cand_xy;   // [M][nobj_total]  (rows written only for valid candidates)
    int* blk_cnt;      // [ceil(M / sample_compact_tile())] valid candidates per compaction tile (zeroed by phase 0, counted by phase 2)
    int* cand_of;      // [M] compacted candidate indices (ascending)
    int* layout_size;  // [1]
    int* n_surv;       // [2] phase-1 survivors, phase-0 survivors
    uint32_t* surv;    // [M][32] phase-1 survivor records
    uint32_t* surv0;   // [M][8] phase-0 survivor records
    float* fake;       // null, or [M][NQ+NV+4]: reset_done's fake step (engine.py:719-724) from rest at the robot
                       // position of valid layout c (row c of the compacted list): qpos, qvel, pose -- robots that move at rest
};

struct DevBuffers {
    float4* dyn;   // [NDYN][Npad] planes of (qpos, qvel, pose0 = (px,py,cos,sin), done0, steps); Point: 3 planes
    float4* obj;   // [P][Npad]: object pairs (goal,h0) (h1,h2) ...
    float4* hist;  // [Npad]: (p1x,p1y,done1,0)   (only when hist_on)
    int* rd_j;     // [Npad]: layout row of a speculated reset_done (see RolloutArgs::do_reset == 2)
    Pool pool;     // the pool the envs are currently drawn from
};

void launch_step(const Params& p, const DevBuffers& b, const float* act, float* obs, float* rew,
                 float* cost, float* done, float* qacc, hipStream_t s);
// `after_phase1`: null, or an event recorded on `s` once the fully VALU-bound phases 0 and 1 are done
int sample_compact_tile(); // candidates per block of the ordered compaction: cand_ok is padded to a multiple of it
// returns the status of the event record it enqueues (ordering-critical: never dropped)
hipError_t launch_sample(const SampleParams& sp, const Pool& pl, hipStream_t s, hipEvent_t after_phase1 = nullptr);
// sharded layout sampling: a shard's valid layouts out (candidate order), the gathered shards in as the pool
// `hdr`: null, or the 4-word header of a piggy-backed export block whose words 1..3 become (k0, k1, tag)
void launch_pool_export(const Pool& pl, int nobj_total, float2* rows, int cap, int* count, hipStream_t s,
                        uint32_t* hdr = nullptr, uint32_t k0 = 0, uint32_t k1 = 0, uint32_t tag = 0);
void launch_pool_install(const Pool& pl, int nobj_total, int n_shards, int cap, const float2* rows_all, const int* counts,
                         int M, hipStream_t s);
// the same from n_shards export blocks [count, k0, k1, shard | n_shards << 16 | rows] `stride_floats` apart (the tails of
// the all-gathered tape shards); blocks sampled for another key / shard / world size are refused (layout_size < 0)
void launch_pool_install_blocks(const Pool& pl, int nobj_total, int n_shards, int cap, const float* blocks,
                                long long stride_floats, uint32_t k0, uint32_t k1, int M, hipStream_t s);
void launch_reset_apply(const Params& p, const DevBuffers& b, int nobj_total, uint32_t k10,
                        uint32_t k11, uint32_t k20, uint32_t k21, float* obs, int* host_layout_size,
                        hipStream_t s);
void launch_reset_done(const Params& p, const DevBuffers& b, int nobj_total, uint32_t k10,
                       uint32_t k11, uint32_t k20, uint32_t k21, const float* obs_in,
                       float* obs_out, hipStream_t s);
// lane-group persistent rollout (small batches): arguments of one launch
struct RolloutArgs {
    int T, nobj_total, hist0;
    int do_reset;      // 0: step only; 1: step + reset_done folded in (rollouts); 2: step, plus the reset_done
                       //    observation of the finished envs SPECULATIVELY into obs_rd and the layout row that
                       //    reset_done would install into rd_j -- the state is not touched (Engine.step)
    int commit;        // apply the pending reset_done recorded in rd_j by the previous (do_reset == 2) launch
    int obs_stride;    // floats between consecutive obs rows (D, or the packed row width D + A + 3)
    int sc_stride;     // floats between consecutive reward / cost / done entries (1, or the packed row width)
    const float* act;
    float* obs;
    float* rew;
    float* cost;
    float* done;
    float* qacc;
    float* act_out;    // null, or where the action of (t, env) is copied: act_out + (t*N + env) * obs_stride
    float* obs_rd;     // do_reset == 2: [N][D] post-reset_done observation
    int* rd_j;         // [Npad] layout row a pending reset_done installs (-1: none)
    const uint4* keys; // per-step (k1, k2) = split(key_t) of the reset_done draw; null: key0 for every step (T == 1)
    uint4 key0;
    const int* layout_size;
    const int* cand_of;
    const float2* cand_xy;
    unsigned long long* stamps; // null, or [grid][8] s_memtime stamps of wave 0 of every workgroup (profiling aid)
    int n_rows;        // rows of cand_xy (layout candidates): bound for every row index read back from memory
    const float* fake; // Pool::fake of the pool in effect (lane-group kernels of the robots that move at rest)
};
void launch_group_rollout(const Params& p, const RolloutArgs& r, const DevBuffers& b, hipStream_t s);
// thread-per-env persistent rollout (large batches)
void launch_thread_rollout(const Params& p, const RolloutArgs& r, const DevBuffers& b, hipStream_t s);
void launch_policy_rollout(const Params& p, const RolloutArgs& r, const PolicyArgs& pol, const DevBuffers& b,
                           int impl, hipStream_t s);
// two-kernel rollout of the light robots (gx_split_rollout.inl): dynamics tape, then one thread per (step, env) row
struct SplitArgs;
bool split_rollout_supported(const Params& p);
int split_tape_width(const Params& p);   // floats per (step, env) tape row: qpos | qvel | action | done | layout rows
int split_entry_width(const Params& p);  // floats per env of the entry record (state before the first step)
// `hold`: null, or an event the observation pass (not the dynamics pass) waits for
// `which`: 3 both passes (gx_rollout), 1 the dynamics pass only (gx_rollout_tape), 2 the observation pass only
// (gx_expand_tape); `entry`: [N][split_entry_width] written by the dynamics pass, read by the observation pass
// returns the status of the stream-ordering calls it makes (the kernels' own launch errors surface in hipGetLastError)
// `lanes`: lanes per env in the dynamics pass where the robot has both forms (Swimmer: 4 = the quad form, faster alone;
// 1 = faster beside a running layout sampler)
// `n_shards` > 1 (observation pass only): one launch over n_shards shard buffers `shard_stride` floats apart (tape, obj0
// and entry all move by it), packed outputs `out_stride` floats apart
hipError_t launch_split_rollout(const Params& p, const RolloutArgs& r, float* tape, float4* obj0, float* entry,
                                const DevBuffers& b, hipStream_t s, hipEvent_t hold = nullptr, int which = 3,
                                int lanes = 1, int n_shards = 1, long long shard_stride = 0, long long out_stride = 0);
// install the reset_done recorded in b.rd_j (pending commit) for consumers other than the lane-group kernels
void launch_commit_pending(const Params& p, const DevBuffers& b, int nobj_total, int n_rows, hipStream_t s);
// fill Pool::fake for the valid layouts of a freshly sampled pool (no-op for robots whose rest state is a fixed point)
void launch_fake_table(const Params& p, const Pool& pl, int nobj_total, int M, hipStream_t s);
int fake_table_width(const Params& p); // floats per row, 0 when the robot needs none
// per-robot launchers: defined in gx_robot_kernels.inl, instantiated once per robot in gx_kernels_<robot>.hip
template <class R>
struct RobotLaunch {
    static void step(const Params& p, const DevBuffers& b, const float* act, float* obs, float* rew, float* cost,
                     float* done, float* qacc, hipStream_t s);
    static void reset_apply(const Params& p, const DevBuffers& b, int nobj_total, uint32_t k10, uint32_t k11,
                            uint32_t k20, uint32_t k21, float* obs, int* host_layout_size, hipStream_t s);
    static void reset_done(const Params& p, const DevBuffers& b, int nobj_total, uint32_t k10, uint32_t k11,
                           uint32_t k20, uint32_t k21, const float* obs_in, float* obs_out, hipStream_t s);
    static void group(const Params& p, const RolloutArgs& r, const DevBuffers& b, hipStream_t s);
    static void thread_rollout(const Params& p, const RolloutArgs& r, const DevBuffers& b, hipStream_t s);
    static void policy(const Params& p, const RolloutArgs& r, const PolicyArgs& pol, const DevBuffers& b, int impl,
                       hipStream_t s);
    static void commit_pending(const Params& p, const DevBuffers& b, int nobj_total, int n_rows, hipStream_t s);
    static void fake_table(const Params& p, const Pool& pl, int nobj_total, int M, hipStream_t s);
    static hipError_t split(const Params& p, const RolloutArgs& r, float* tape, float4* obj0, float* entry,
                            const DevBuffers& b, hipStream_t s, hipEvent_t hold, int which, int lanes, int n_shards,
                            long long shard_stride, long long out_stride);
    static int split_width();
    static int split_entry_width();
};
// step-wise policy (gx_policy_step.hip): hidden widths whose weights do not fit the LDS of the fused kernel
bool policy_step_supported(int H);
int policy_step_wt_floats(int D, int H);
void launch_policy_transpose(const float* params, float* wt, int D, int A, int H, hipStream_t s);
// valu: sequential fmaf chains on the vector ALUs instead of v_mfma_f32_16x16x4_f32 tiles (same bits)
// mode 0: ac.step for one time step (obs -> obs_in, act, mu, logp, val); mode 1: critic only (val -> val_last, obs_last)
void launch_policy_step(int H, const float* params, const float* wt, const float* obs, uint32_t seed0, uint32_t seed1,
                        uint32_t tnoise, int N, int D, int A, int env_offset, int mode, float* obs_in, float* act, float* mu,
                        float* logp, float* val, float* obs_last, float* logstd, hipStream_t s, bool valu = false);
bool policy_rollout_supported(const Params& p);
bool policy_fused128_supported(const Params& p);
size_t policy_lds_bytes(const Params& p, int impl);
void launch_math_probe2(int n, const float* x, float* lg, float* th, hipStream_t s);
void launch_math_probe(int n, const float* x, const float* y, float* s_, float* c, float* at2,
                       float* ex, hipStream_t s);
void launch_split_probe(uint32_t k0, uint32_t k1, int n, uint32_t* out, hipStream_t s);
size_t step_lds_bytes(const Params& p, int block);
int pick_block(const Params& p);

} // namespace gx
