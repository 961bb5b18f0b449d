// gx_split_rollout.inl -- the fused T-step rollout at small env_num as TWO kernels instead of one persistent lane-group
// kernel (included by gx_robot_kernels.inl):
//
//   pass 1  the serial chain: convert_action, mjx.step, done / NaN guard / timeout, reset_done (layout index draw +
//           re-placement) -- everything the NEXT step depends on -- and one SLIM tape row per (step, env): qpos, qvel after
//           the step, the action, done, the layout row in effect and the layout row a reset_done installed.
//           dyn_tape_kernel (Point, Swimmer): one thread per env, 9 / 13 floats per row, ~170 instructions per Point step
//           (the Swimmer also as a quad of lanes per env, SwimmerRobot::substep_q).
//           group_dyn_tape_kernel (Ant, Walker; round 3): the lane-group form of their step, 16 lanes per env, 32 / 38
//           floats per row (incl. the row of the pool's fake-step table a reset_done observation is read from).
//           (Rounds 1-2 also wrote the stepped pose, ctrl and the reward into the row.)
//   pass 2  obs_tape_kernel   one thread per (step, env) tape row: re-derives what pass 1 no longer writes -- the pose
//           the step returned (kinematics of the qpos the step STARTED from: the previous row's qpos, or the robot
//           position of the layout the previous row's reset_done installed), the pose before that (for
//           convert_action -> ctrl and for reward_done's `last`), the reward -- then lidars, compass, cost, the
//           observation row (of the re-initialised env where reset_done fired) and the reward / cost / done outputs:
//           400 000 independent rows at env_num = 2000, T = 200, the bandwidth regime of the thread-per-env kernels.
//           Rows 0 and 1 of an env take the state at entry from the entry record pass 1 leaves per env.
//
// Same functions, same operation order as step_kernel / reset_done_kernel, hence the same bits
// (tests/test_gpu_parity.py runs every rollout test on this path too).  Not used when observe_vel / observe_acc
// need the pose history in the row, with more than one physics step per control step, for the closed-loop policy
// rollout, or for Engine.step.
//
// The NaN guard (engine.py:696-699) needs "any observation entry non-finite" in pass 1.  With finite qpos / qvel /
// ctrl / pose, |position| < 1e18 and finite objects of that size every entry is finite (exp <= 1, alias in [0,1],
// compass a sum of two products < 1e37), so that test decides almost every step; whenever it does not hold, pass 1
// evaluates the observation exactly (build_obs_row into an LDS row) like the one-kernel paths.
#pragma once
#include <type_traits>

namespace gx {

template <class R>
struct SplitTape {
    // One row per (step, env): qpos | qvel after the step | the action | kCode: ONE word for done, the layout row in effect
    // and the layout row a reset_done installed | kFidx (Ant, Walker): row of Pool::fake (= index into the compacted layout
    // list) of the installed layout.
    //   kCode >= 0: the step did not finish the env; the layout in effect is row kCode - 1 of the pool (0: the layout at
    //               entry);
    //   kCode == -1: it did, and no layout was installed (no reset_done in this launch, or an empty pool);
    //   kCode <= -2: it did, and reset_done installed layout row -(kCode + 2).
    // The layout in effect during a step that DID finish the env is what the rows before it say (jcur_before): the row
    // in front of it, almost always -- pass 2 has loaded that one anyway.
    // Rounds 3-4 carried `done`, the layout in effect and the installed one in words of their own and padded the row to
    // 16, then 8 bytes (Point: 12, then 10 floats); now 9 floats = 36 B (Swimmer 13, Ant 32, Walker 38) -- this is what the
    // multi-GPU hand-off puts on the wire, 400 000 rows per rank and epoch, and at W = 8 the link is the bound.  Rows are
    // 4-byte aligned only.
    static constexpr int kQ = 0, kV = kQ + R::NQ, kAct = kV + R::NV, kCode = kAct + R::NA,
                         kFidx = kCode + 1, kUsed = kFidx + (R::kRestFixed ? 0 : 1), kW = kUsed;
    // entry record of an env: qpos at entry | the stale pose (x, y, cos, sin) | done0 | number of step() calls so far
    static constexpr int kEQ = 0, kEPose = R::NQ, kEDone = kEPose + 4, kEHist = kEDone + 1, kE = (kEHist + 1 + 3) / 4 * 4;
    static_assert(!R::kRestFixed || kE == 12, "entry record of the light robots: 12 floats (include/guardx.h)");
    GX_D static int code(int jcur, float dn, int jaft) { return dn > 0.0f ? (jaft >= 0 ? -(jaft + 2) : -1) : jcur + 1; }
    GX_D static float done_of(int c) { return c < 0 ? 1.0f : 0.0f; }
    GX_D static int jaft_of(int c) { return c <= -2 ? -(c + 2) : -1; }
    // the layout row in effect when step t of the env behind tape row gg starts (-1: the layout at entry), read off the
    // rows before it: the last one that names a layout (a finished step without an install names none)
    GX_D static int jcur_before(const float* __restrict__ tape, size_t gg, size_t N, int t)
    {
        for (int k = 1; k <= t; ++k) {
            const int c = __float_as_int(tape[(gg - (size_t)k * N) * kW + kCode]);
            if (c >= 0) return c - 1;
            if (c <= -2) return -(c + 2);
        }
        return -1;
    }
};

struct SplitArgs {
    float* tape;        // [T][N][kW]
    float4* obj0;       // [P][Npad] snapshot of the layouts at entry (pass 2 reads it for rows with jcur < 0)
    float* entry;       // [N][kE] state at entry (pass 2 needs it for the rows of steps 0 and 1)
    int lanes;          // lanes per env in pass 1 where the robot offers a choice (R::kDynLanes): 1 or 4
    // pass 2 over several shards in ONE launch (gx_expand_tapes; blockIdx.y = shard): floats between the buffers of
    // consecutive shards (tape, obj0 and entry all move by it) and between their packed outputs
    long long shard_stride, out_stride;
};

GX_D bool moderate(float x) { return fabsf(x) < 1e18f; } // false for NaN / Inf too
// the dynamics pass's bound on the sum of |qpos|, |qvel| (and |action|) of a state a common step may start from: 2^24,
// so that every angle in it is within the range the unguarded sincos_f<true> reduces exactly like the guarded one
GX_D bool state_ok(float sum) { return fabsf(sum) < 16777216.0f; }
constexpr int kActBlock = 16; // steps whose actions the dynamics pass fetches at once
constexpr int kObsGridCap = 3072; // one-wave workgroups of a ONE-shard observation launch (measured: 47.1 -> 44.2 us at 400 000 rows; a launch over
                                  // several shards is fastest uncapped: 32.5 us per shard at 8 shards)

// rows of W floats: 16-byte pieces, then an 8-byte and / or a 4-byte one.  Tape rows start on 4-byte boundaries only
// (kW = 9 for the Point), so the wide accesses are declared with 4-byte alignment (the hardware takes a dwordx4 at any
// dword address; the compiler must not be told more than is true); entry records are 16-byte aligned rows of 12.
typedef float gx_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float gx_f2u __attribute__((ext_vector_type(2), aligned(4)));
template <int W>
GX_D void load_row(const float* __restrict__ p, float (&v)[W])
{
#pragma unroll
    for (int k = 0; k < W / 4; ++k) {
        const gx_f4u t4 = *reinterpret_cast<const gx_f4u*>(p + 4 * k);
        v[4 * k] = t4.x; v[4 * k + 1] = t4.y; v[4 * k + 2] = t4.z; v[4 * k + 3] = t4.w;
    }
    constexpr int B = W / 4 * 4;
    if (W % 4 >= 2) {
        const gx_f2u t2 = *reinterpret_cast<const gx_f2u*>(p + B);
        v[B] = t2.x; v[B + 1] = t2.y;
    }
    if (W % 2) v[W - 1] = p[W - 1];
}
template <int W>
GX_D void store_row(float* __restrict__ p, const float (&v)[W])
{
#pragma unroll
    for (int k = 0; k < W / 4; ++k) {
        gx_f4u t4; t4.x = v[4 * k]; t4.y = v[4 * k + 1]; t4.z = v[4 * k + 2]; t4.w = v[4 * k + 3];
        *reinterpret_cast<gx_f4u*>(p + 4 * k) = t4;
    }
    constexpr int B = W / 4 * 4;
    if (W % 4 >= 2) {
        gx_f2u t2; t2.x = v[B]; t2.y = v[B + 1];
        *reinterpret_cast<gx_f2u*>(p + B) = t2;
    }
    if (W % 2) p[W - 1] = v[W - 1];
}

template <class R, int BLOCK, int PMAX, bool kDef, int LPE = 1>
__global__ __launch_bounds__(BLOCK) void dyn_tape_kernel(Params p_in, RolloutArgs r, SplitArgs sa,
                                                         float4* __restrict__ dyn, float4* __restrict__ obj)
{
    using TP = SplitTape<R>;
    // the serial chain of the epoch: its waves go first wherever they share a SIMD with the layout sampler's (which are
    // throughput work and fill every issue slot this wave leaves)
    __builtin_amdgcn_s_setprio(3);
    const Params p = fold_params<R, kDef>(p_in);
    extern __shared__ float4 tile4[];
    float* tile = reinterpret_cast<float*>(tile4); // one obs row per thread: exact NaN-guard evaluation only
    // LPE lanes per env (4, Swimmer: the lanes of a quad share an env and split the step's independent pieces,
    // SwimmerRobot::substep_q; they all carry the state, lane 0 of the quad does the stores)
    const int tid = threadIdx.x;
    const int i = (blockIdx.x * BLOCK + tid) / LPE;
    const int jq = tid & (LPE - 1);
    const bool writer = jq == 0;
    if (i >= p.N) return;
    float q[R::NQ], v[R::NV], pose0[4], done0, steps;
    R::load(dyn, p.Npad, i, q, v, pose0, done0, steps);
    if (writer) {   // the state at entry, for the rows of steps 0 and 1 in pass 2
        float ev[TP::kE];
#pragma unroll
        for (int k = 0; k < TP::kE; ++k) ev[k] = 0.f;
#pragma unroll
        for (int k = 0; k < R::NQ; ++k) ev[TP::kEQ + k] = q[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) ev[TP::kEPose + k] = pose0[k];
        ev[TP::kEDone] = done0; ev[TP::kEHist] = (float)r.hist0;
        store_row<TP::kE>(sa.entry + (size_t)i * TP::kE, ev);
    }
    // layout at entry: snapshot for pass 2, goal for done, magnitude check for the NaN-guard shortcut
    float gx = 0.f, gy = 0.f;
    // the shortcut also needs a closeness that cannot overflow: exp(-gain*dist) with gain >= 0, or a positive max_dist
    const bool cfg_ok = p.lidar_max_dist_set ? (p.lidar_max_dist > 0.0f) : (p.neg_gain <= 0.0f);
    bool objs_ok = cfg_ok;
#pragma unroll
    for (int k = 0; k < PMAX; ++k) {
        if (k < p.P) {
            const float4 o4 = obj[(size_t)k * p.Npad + i];
            if (writer) sa.obj0[(size_t)k * p.Npad + i] = o4;
            if (k == 0) { gx = o4.x; gy = o4.y; }
            objs_ok = objs_ok && moderate(o4.x) && moderate(o4.y);
            if (2 * k + 1 < p.nobj) objs_ok = objs_ok && moderate(o4.z) && moderate(o4.w);
        }
    }
    const int L = r.do_reset ? *r.layout_size : 0;
    int jcur = -1;
    float* row = tile + tid * p.D;
    // ACTIONS THROUGH LDS.  vmcnt counts loads and stores alike, in order: a per-step action load, however early it is
    // issued, makes the step that consumes it wait (s_waitcnt vmcnt) for the tape stores issued after it too -- one L2
    // store round trip per step, which is what bounded this kernel in rounds 1-2 (~580 ns per step whatever the
    // instruction count).  The actions of kActBlock steps are therefore fetched at once into registers a whole block
    // ahead, parked in LDS (lgkmcnt: its own counter) when the block ends, and read from there step by step: the loop
    // waits for global memory once per kActBlock steps, and the tape stores drain behind the arithmetic.
    float* actl = tile + BLOCK * p.D; // [2][kActBlock][BLOCK][NA]
    float anx[kActBlock][R::NA];
#pragma unroll
    for (int k = 0; k < kActBlock; ++k) {
#pragma unroll
        for (int d = 0; d < R::NA; ++d) anx[k][d] = 0.f;
        if (k < r.T) load_action<R>(r.act, (size_t)k * p.N + i, anx[k]);
    }
#pragma unroll
    for (int k = 0; k < kActBlock; ++k)
#pragma unroll
        for (int d = 0; d < R::NA; ++d) actl[(k * BLOCK + tid) * R::NA + d] = anx[k][d];
    int abuf = 0;
    // FAST PATH / EXACT PATH.  Almost every step is "ordinary": nothing is NaN or huge and the robot moved a few
    // centimetres.  For such a step (i) jp.clip is the bare median (no NaN to put back), (ii) no observation entry can be
    // non-finite (exp <= 1, alias in [0,1], compass a sum of two products < 1e37), (iii) reward_done's teleport test
    // |last - dist| > 1 is false by the triangle inequality (both distances are to the same goal, from positions less
    // than 0.95 apart; their correctly rounded square roots differ from the true ones by < 1e-3 below 1e4), and
    // (iv) dist < goal_size  <=>  dist^2 < goal_cut exactly -- so the step needs no NaN select, no observation and NO
    // SQUARE ROOT (the reward is the observation pass's).  Whether the step was ordinary is checked AFTER it (one sum of
    // magnitudes, two squared lengths); if not, the step is redone from the saved state with the exact forms: same
    // results either way, the serial chain is ~45 instructions shorter.
    //   s_ok: qpos, qvel at the start of the step are finite and < 1e18 (so is the pose the step returns, and its
    //         velocity-servo term cannot be NaN);  p_ok: the stale pose's cos / sin are finite (ctrl = pose0 * action)
    // (round 4: one flag for both -- a common step starts with it set and leaves it set, so the serial chain carries
    // no flag updates at all; only the general step below recomputes it)
    // WAVE MASKS, not per-lane bools: the flags live as 64-bit lane masks in scalar registers (one ballot each), the
    // tests of a step are ballots ANDed on the scalar unit, and one scalar compare decides "every lane had a common
    // step".  (As per-lane bools the compiler kept merging them under the exec mask at the end of every step: ~20
    // scalar and vector instructions on the serial chain for flags that almost never change.)
    typedef unsigned long long mask_t;
    const mask_t live_m = __builtin_amdgcn_ballot_w64(true);
    const int lane_id = tid & 63;
    mask_t objs_m = __builtin_amdgcn_ballot_w64(objs_ok), sp_m;
    {
        float m0 = fabsf(pose0[2]) + fabsf(pose0[3]), m1 = 0.f;
#pragma unroll
        for (int k = 0; k < R::NQ; ++k) m1 = m1 + fabsf(q[k]);
#pragma unroll
        for (int k = 0; k < R::NV; ++k) m1 = m1 + fabsf(v[k]);
        sp_m = __builtin_amdgcn_ballot_w64(state_ok(m1)) & __builtin_amdgcn_ballot_w64(moderate(m0));
    }
    const mask_t phys1_m = p.physics_steps == 1 ? ~0ull : 0ull;
#pragma unroll 1
    for (int tb = 0; tb < r.T; tb += kActBlock) { // blocks of kActBlock steps (two loops: the block's addresses are
                                                  // computed once per block, not carried through every step)
    if (tb + kActBlock < r.T) { // request the next block's actions (consumed kActBlock steps from now)
#pragma unroll
        for (int k = 0; k < kActBlock; ++k)
            if (tb + kActBlock + k < r.T) load_action<R>(r.act, (size_t)(tb + kActBlock + k) * p.N + i, anx[k]);
    }
    const int kend = r.T - tb < kActBlock ? r.T - tb : kActBlock;
    // (two steps per trip: the stepped state of the first is the start of the second in place -- a one-step loop copies
    // every loop-carried register back at the end of each step, ~30 moves on the serial chain)
    auto step1 = [&](const int kb) __attribute__((always_inline)) {
        const int t = tb + kb;
        float a[R::NA];
#pragma unroll
        for (int d = 0; d < R::NA; ++d) a[d] = actl[((abuf * kActBlock + kb) * BLOCK + tid) * R::NA + d];
        const bool have_last = (r.hist0 + t) >= 1;
        const float last_done = done0;

        float ctrl[R::NU];
        R::convert_action(pose0, a, ctrl); // :672-685, PRE-step xmat
        float pose[4], qacc[R::NV], qf[R::NQ], vf[R::NV];
#pragma unroll
        for (int k = 0; k < R::NQ; ++k) qf[k] = q[k];
#pragma unroll
        for (int k = 0; k < R::NV; ++k) vf[k] = v[k];
        if constexpr (LPE == 4) R::template substep_q<true>(qf, vf, ctrl, pose, qacc, jq);
        else R::template substep<false, true>(qf, vf, ctrl, pose, qacc);
        world_pose(p, pose);
        // sum of magnitudes of the stepped state and the action, as a tree (it only feeds a bound: any order will do,
        // and the chain from the last velocity to the branch is three additions long instead of eight)
        float mag;
        {
            float term[R::NQ + R::NV + R::NA];
#pragma unroll
            for (int k = 0; k < R::NQ; ++k) term[k] = fabsf(qf[k]);
#pragma unroll
            for (int k = 0; k < R::NV; ++k) term[R::NQ + k] = fabsf(vf[k]);
#pragma unroll
            for (int k = 0; k < R::NA; ++k) term[R::NQ + R::NV + k] = fabsf(a[k]);
            constexpr int n = R::NQ + R::NV + R::NA;
#pragma unroll
            for (int w = 1; w < n; w *= 2)
#pragma unroll
                for (int k = 0; k + w < n; k += 2 * w) term[k] = term[k] + term[k + w];
            mag = term[0];
        }
        const float mx = pose[0] - pose0[0], my = pose[1] - pose0[1];
        const float gdx = gx - pose[0], gdy = gy - pose[1];
        const float d2 = gdx * gdx + gdy * gdy;                 // dist2()'s radicand
        // (five compares into scalar lane masks and scalar ANDs)
        const mask_t ord_m = objs_m & sp_m & phys1_m & __builtin_amdgcn_ballot_w64(state_ok(mag)) &
                             __builtin_amdgcn_ballot_w64((mx * mx + my * my) < 0.9f) & __builtin_amdgcn_ballot_w64(d2 < 1e8f);
        // COMMON STEP vs GENERAL STEP (round 4).  Almost every step of almost every wave is ordinary, finishes no env
        // (no goal reached, no timeout) and so re-initialises nothing.  The per-lane branches for the other cases -- the
        // exact redo, done, the reset_done draw, the re-placement -- cost this serial chain an exec-mask save / restore
        // and a taken jump over kilobytes of cold code each, every step (stubbing the dynamics showed 86 of the Point's
        // 114 us per 200 steps in this wrapper, not in the step).  One wave-uniform test now selects a branch-free
        // commit; any lane that needs more sends the whole wave through the general code below (same results).
        const mask_t rare_m = (~ord_m & live_m) | __builtin_amdgcn_ballot_w64(d2 < p.goal_cut) |
                              __builtin_amdgcn_ballot_w64(steps > p.num_steps_f);
        if (__builtin_expect(rare_m == 0ull, 1)) {
#pragma unroll
            for (int k = 0; k < R::NQ; ++k) q[k] = qf[k];
#pragma unroll
            for (int k = 0; k < R::NV; ++k) v[k] = vf[k];
            steps = steps + 1.0f;                       // :493 (done == 0); the flag masks stay as they are
            float rowv[TP::kW];
#pragma unroll
            for (int k = 0; k < R::NQ; ++k) rowv[TP::kQ + k] = q[k];
#pragma unroll
            for (int k = 0; k < R::NV; ++k) rowv[TP::kV + k] = v[k];
#pragma unroll
            for (int k = 0; k < R::NA; ++k) rowv[TP::kAct + k] = a[k];
            rowv[TP::kCode] = __int_as_float(TP::code(jcur, 0.0f, -1));
#pragma unroll
            for (int k = TP::kUsed; k < TP::kW; ++k) rowv[k] = 0.f;
            if (writer) store_row<TP::kW>(sa.tape + ((size_t)t * p.N + i) * TP::kW, rowv);
#pragma unroll
            for (int k = 0; k < 4; ++k) pose0[k] = pose[k];
            done0 = 0.0f;
        } else {
            const bool ordinary = (ord_m >> lane_id) & 1ull;
            bool sp_ok;
            objs_ok = (objs_m >> lane_id) & 1ull;
            float dn;
            if (ordinary) {
#pragma unroll
                for (int k = 0; k < R::NQ; ++k) q[k] = qf[k];
#pragma unroll
                for (int k = 0; k < R::NV; ++k) v[k] = vf[k];
                dn = d2 < p.goal_cut ? 1.0f : 0.0f;
                sp_ok = true;  // this step's pose: the kinematics of a moderate qpos; the state: moderate(mag)
            } else { // rare: the step again, exactly (from the untouched q, v)
                for (int k = 0; k < p.physics_steps; ++k) {
                    if constexpr (LPE == 4) R::template substep_q<true>(q, v, ctrl, pose, qacc, jq);
                    else R::template substep<false>(q, v, ctrl, pose, qacc);
                }
                world_pose(p, pose);
                // NaN / Inf guard :696-699
                float4 ob[PMAX];
                if (jcur >= 0) { float rx_, ry_; load_layout<PMAX>(p, r.cand_xy, r.nobj_total, jcur, ob, rx_, ry_); }
                else {
#pragma unroll
                    for (int k = 0; k < PMAX; ++k)
                        ob[k] = (k < p.P) ? sa.obj0[(size_t)k * p.Npad + i] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                const bool bad = build_obs_row<R, PMAX>(p, row, pose, ob, ctrl, q, v, 0.f, 0.f, 0.f, 0.f);
                // the done half of reward_done :787-802 (the reward itself is pass 2's)
                const float dg = dist2(gx, gy, pose[0], pose[1]);
                float last = dg;
                if (have_last && !(last_done > 0.0f)) last = dist2(gx, gy, pose0[0], pose0[1]);
                const float dd = last - dg;
                dn = dg < p.goal_size ? 1.0f : 0.0f;
                if (fabsf(dd) > 1.0f) dn = 1.0f;
                if (bad) dn = 1.0f;                    // :696-699
                float m1 = 0.f;
#pragma unroll
                for (int k = 0; k < R::NQ; ++k) m1 = m1 + fabsf(q[k]);
#pragma unroll
                for (int k = 0; k < R::NV; ++k) m1 = m1 + fabsf(v[k]);
                sp_ok = (int)moderate(fabsf(pose[2]) + fabsf(pose[3])) & (int)state_ok(m1);
            }
            if (steps > p.num_steps_f) dn = 1.0f;      // :492
            steps = dn > 0.0f ? 0.0f : steps + 1.0f;   // :493

            // reset_done :497-505 for the env that just finished: the draw and the re-placement
            int jaft = -1;
            float nq0 = 0.f, nq1 = 0.f;
            if (r.do_reset && dn > 0.0f && L > 0) {
                const uint4 kk = r.keys ? r.keys[t] : r.key0;
                const uint32_t idx = randint_at(kk.x, kk.y, kk.z, kk.w, (uint32_t)p.env_total, (uint32_t)L,
                                                (uint32_t)(p.env_offset + i));
                jaft = r.cand_of[idx];
                const float2* rowp = r.cand_xy + (size_t)jaft * r.nobj_total;
                const float2 g = rowp[0], rb = rowp[r.nobj_total - 1];
                nq0 = rb.x; nq1 = rb.y;
                gx = g.x; gy = g.y;
                // The loads of this RARE branch must have landed before it ends: otherwise the compiler guards the next
                // step's first touch of these registers with an s_waitcnt vmcnt(3) on the COMMON path -- and the memory
                // counter is in order, so that wait also covers the tape stores of the step before the previous one: a
                // store round trip on the serial chain of almost every step, for a load that almost never happened.
                asm volatile("" : "+v"(nq0), "+v"(nq1), "+v"(gx), "+v"(gy));
            }

            // tape row
            float rowv[TP::kW];
#pragma unroll
            for (int k = 0; k < R::NQ; ++k) rowv[TP::kQ + k] = q[k];
#pragma unroll
            for (int k = 0; k < R::NV; ++k) rowv[TP::kV + k] = v[k];
#pragma unroll
            for (int k = 0; k < R::NA; ++k) rowv[TP::kAct + k] = a[k];
            rowv[TP::kCode] = __int_as_float(TP::code(jcur, dn, jaft));
#pragma unroll
            for (int k = TP::kUsed; k < TP::kW; ++k) rowv[k] = 0.f;
            if (writer) store_row<TP::kW>(sa.tape + ((size_t)t * p.N + i) * TP::kW, rowv);

            // commit the history, then the re-initialisation (the stale pose stays, :731)
#pragma unroll
            for (int k = 0; k < 4; ++k) pose0[k] = pose[k];
            done0 = dn;
            if (jaft >= 0) {
#pragma unroll
                for (int k = 0; k < R::NQ; ++k) q[k] = 0.f;
#pragma unroll
                for (int k = 0; k < R::NV; ++k) v[k] = 0.f;
                R::place(q, nq0, nq1);
                jcur = jaft;
                objs_ok = cfg_ok; // pool rows lie inside the placement extents
                // ... and so does the robot, at rest; the stale pose is the one just checked
                sp_ok = (int)moderate(fabsf(pose[2]) + fabsf(pose[3])) & (int)state_ok(fabsf(nq0) + fabsf(nq1));
            }
            objs_m = __builtin_amdgcn_ballot_w64(objs_ok);
            sp_m = __builtin_amdgcn_ballot_w64(sp_ok);
        }
    };
    int kb = 0;
#pragma unroll 1
    for (; kb + 1 < kend; kb += 2) { step1(kb); step1(kb + 1); }
    if (kb < kend) step1(kb);
    abuf ^= 1; // park the next block's actions
#pragma unroll
    for (int k = 0; k < kActBlock; ++k)
#pragma unroll
        for (int d = 0; d < R::NA; ++d) actl[((abuf * kActBlock + k) * BLOCK + tid) * R::NA + d] = anx[k][d];
    }
    if (writer) R::store(dyn, p.Npad, i, q, v, pose0, done0, steps);
    if (writer && jcur >= 0) { // the layout a reset_done installed becomes the env's layout
        float4 ob[PMAX];
        float rx_, ry_;
        load_layout<PMAX>(p, r.cand_xy, r.nobj_total, jcur, ob, rx_, ry_);
#pragma unroll
        for (int k = 0; k < PMAX; ++k)
            if (k < p.P) obj[(size_t)k * p.Npad + i] = ob[k];
    }
}

// ---------------------------------------------------------------------------
// Pass 1 for the robots with contact dynamics (Ant, Walker): the lane-group form of the step (16 lanes per env, rows /
// bodies / right-hand sides of a leg spread over the lanes: gx_robot_ant_group.h, gx_robot_legs_group.h) without the
// observation.  Per step the persistent lane-group rollout kernel spends ~1.4 us on the lidar exchange and the rows and
// waits once for its own stores (every __syncthreads of the exchange is also an s_waitcnt vmcnt(0)); here a step is
// the dynamics, two square roots and one 144 / 160-byte tape row written by the group's first lane, with the actions of
// kActBlock steps parked in LDS (see dyn_tape_kernel).  The NaN guard takes the same shortcut: finite, moderate qpos /
// qvel / action and moderate objects cannot produce a non-finite observation entry (ctrl is the action, the pose the
// kinematics of a moderate qpos); otherwise the wave evaluates the observation exactly with group_observe.
// ---------------------------------------------------------------------------
template <class R, int OPL, int BPL, bool kDef>
__global__ __launch_bounds__(64) void group_dyn_tape_kernel(Params p_in, RolloutArgs r, SplitArgs sa,
                                                            float4* __restrict__ dyn, float4* __restrict__ obj)
{
    using TP = SplitTape<R>;
    constexpr int BT = 64, EPW = BT / kGL; // envs per wave
    __builtin_amdgcn_s_setprio(3); // see dyn_tape_kernel
    const Params p = fold_params<R, kDef>(p_in);
    __shared__ GroupLds<OPL, BPL, BT> S;
    __shared__ float actl[2][kActBlock][EPW][R::NA];
    const int lane = threadIdx.x, l = lane & (kGL - 1), g = lane >> 4;
    // envs of this wave: 4, or -- sa.lanes = 1 | 2, chosen by the launcher while the waves still fit one per SIMD --
    // fewer: the Newton loop of the step runs until the LAST env of the wave has converged, and the maximum over one or
    // two envs is smaller than over four.  The other groups of lanes repeat the envs of the first (same data, same trip
    // counts, no stores).
    const int epw = (sa.lanes == 1 || sa.lanes == 2) ? sa.lanes : EPW;
    const int env = blockIdx.x * epw + (g & (epw - 1));
    const bool twin = env < p.N;            // computes (a group beyond epw repeats its twin's env, reset_done included)
    const bool live = twin && g < epw;       // ... and stores
    const int e = twin ? env : 0;
    const bool writer = live && l == 0;

    float q[R::NQ], v[R::NV], pose0[4], done0, steps;
    R::load(dyn, p.Npad, e, q, v, pose0, done0, steps);
    if (writer) { // the state at entry, for the rows of steps 0 and 1 in pass 2
        float ev[TP::kE];
#pragma unroll
        for (int k = 0; k < TP::kE; ++k) ev[k] = 0.f;
#pragma unroll
        for (int k = 0; k < R::NQ; ++k) ev[TP::kEQ + k] = q[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) ev[TP::kEPose + k] = pose0[k];
        ev[TP::kEDone] = done0; ev[TP::kEHist] = (float)r.hist0;
        store_row<TP::kE>(sa.entry + (size_t)env * TP::kE, ev);
    }
    // layout at entry: snapshot for pass 2 (lane k copies pair k), goal, magnitude check for the NaN-guard shortcut
    const float2* obj2 = reinterpret_cast<const float2*>(obj);
    float gx, gy;
    { const float2 g2 = obj2[(size_t)e * 2]; gx = g2.x; gy = g2.y; }
    const bool cfg_ok = p.lidar_max_dist_set ? (p.lidar_max_dist > 0.0f) : (p.neg_gain <= 0.0f);
    bool mine_ok = true;
    for (int k = l; k < p.P; k += kGL) {
        const float4 o4 = obj[(size_t)k * p.Npad + e];
        if (live) sa.obj0[(size_t)k * p.Npad + e] = o4;
        mine_ok = mine_ok && moderate(o4.x) && moderate(o4.y);
        if (2 * k + 1 < p.nobj) mine_ok = mine_ok && moderate(o4.z) && moderate(o4.w);
    }
    const int gsh = (lane & ~(kGL - 1)) & 63;
    bool objs_ok = cfg_ok && (((__ballot(!mine_ok) >> gsh) & 0xFFFFull) == 0ull);
    const int L = r.do_reset ? *r.layout_size : 0;
    int jcur = -1;
    bool s_ok;
    {
        float m1 = 0.f;
#pragma unroll
        for (int k = 0; k < R::NQ; ++k) m1 = m1 + fabsf(q[k]);
#pragma unroll
        for (int k = 0; k < R::NV; ++k) m1 = m1 + fabsf(v[k]);
        s_ok = moderate(m1);
    }
    // actions: lane d of the group fetches entry d of kActBlock steps at once, parks them in LDS a block ahead
    float anx[kActBlock];
#pragma unroll
    for (int k = 0; k < kActBlock; ++k) {
        anx[k] = 0.f;
        if (k < r.T && l < R::NA) anx[k] = r.act[((size_t)k * p.N + e) * R::NA + l];
    }
    if (l < R::NA) {
#pragma unroll
        for (int k = 0; k < kActBlock; ++k) actl[0][k][g][l] = anx[k];
    }
    int abuf = 0;
#pragma unroll 1
    for (int tb = 0; tb < r.T; tb += kActBlock) {
    if (tb + kActBlock < r.T && l < R::NA) {
#pragma unroll
        for (int k = 0; k < kActBlock; ++k)
            if (tb + kActBlock + k < r.T) anx[k] = r.act[((size_t)(tb + kActBlock + k) * p.N + e) * R::NA + l];
    }
    const int kend = r.T - tb < kActBlock ? r.T - tb : kActBlock;
#pragma unroll 1
    for (int kb = 0; kb < kend; ++kb) {
        const int t = tb + kb;
        float a[R::NA];
#pragma unroll
        for (int d = 0; d < R::NA; ++d) a[d] = actl[abuf][kb][g][d];
        const bool have_last = (r.hist0 + t) >= 1;
        const float last_done = done0;
        const float L1x = pose0[0], L1y = pose0[1];

        float ctrl[R::NU];
        R::convert_action(pose0, a, ctrl);
        float pose[4], qacc[R::NV];
#pragma unroll
        for (int k = 0; k < R::NV; ++k) qacc[k] = 0.f;
        group_substep<R, false>(q, v, ctrl, pose, qacc, l);
        world_pose(p, pose);

        // NaN / Inf guard :696-699
        float mag = 0.f;
#pragma unroll
        for (int k = 0; k < R::NQ; ++k) mag = mag + fabsf(q[k]);
#pragma unroll
        for (int k = 0; k < R::NV; ++k) mag = mag + fabsf(v[k]);
        const bool sm_ok = moderate(mag); // the stepped state: the next step's start
#pragma unroll
        for (int k = 0; k < R::NA; ++k) mag = mag + fabsf(a[k]);
        const bool ordinary = objs_ok && s_ok && moderate(mag);
        bool bad = false;
        if (__ballot(!ordinary) != 0ull) { // rare, wave-uniform: the observation exactly
            float ox[OPL], oy[OPL];
#pragma unroll
            for (int j = 0; j < OPL; ++j) {
                const int o = l + kGL * j;
                float2 t2 = make_float2(0.f, 0.f);
                if (o < p.nobj) {
                    if (jcur >= 0) t2 = r.cand_xy[(size_t)jcur * r.nobj_total + o];
                    else t2 = obj2[((size_t)(o >> 1) * p.Npad + e) * 2 + (o & 1)]; // `obj` is rewritten only at the end
                }
                ox[j] = t2.x; oy[j] = t2.y;
            }
            const GroupObs<OPL, BPL> ob = group_observe<OPL, BPL, BT, false>(p, S, lane, pose, gx, gy, ox, oy);
            bad = ob.bad;
            if (p.off_ctrl >= 0) {
#pragma unroll
                for (int k = 0; k < R::NU; ++k) bad = bad || notfinite(ctrl[k]);
            }
            if (p.off_qpos >= 0) {
#pragma unroll
                for (int k = 0; k < R::NQ; ++k) bad = bad || notfinite(q[k]);
            }
            if (p.off_qvel >= 0) {
#pragma unroll
                for (int k = 0; k < R::NV; ++k) bad = bad || notfinite(v[k]);
            }
        }
        // the done half of reward_done :787-802 (the reward itself is pass 2's)
        const float dg = dist2(gx, gy, pose[0], pose[1]);
        float last = dg;
        if (have_last && !(last_done > 0.0f)) last = dist2(gx, gy, L1x, L1y);
        const float dd = last - dg;
        float dn = dg < p.goal_size ? 1.0f : 0.0f;
        if (fabsf(dd) > 1.0f) dn = 1.0f;
        if (bad) dn = 1.0f;                        // :696-699
        if (steps > p.num_steps_f) dn = 1.0f;      // :492
        steps = dn > 0.0f ? 0.0f : steps + 1.0f;   // :493

        // reset_done :497-505 for the env that just finished: the draw and the re-placement
        int jaft = -1, fidx = 0;
        float nq0 = 0.f, nq1 = 0.f;
        if (r.do_reset && twin && dn > 0.0f && L > 0) {
            const uint4 kk = r.keys ? r.keys[t] : r.key0;
            const uint32_t idx = randint_at(kk.x, kk.y, kk.z, kk.w, (uint32_t)p.env_total, (uint32_t)L,
                                            (uint32_t)(p.env_offset + env));
            jaft = r.cand_of[idx];
            fidx = (int)idx;
            const float2* rowp = r.cand_xy + (size_t)jaft * r.nobj_total;
            const float2 g2 = rowp[0], rb = rowp[r.nobj_total - 1];
            nq0 = rb.x; nq1 = rb.y;
            gx = g2.x; gy = g2.y;
        }
        if (writer) { // tape row
            float rowv[TP::kW];
#pragma unroll
            for (int k = 0; k < R::NQ; ++k) rowv[TP::kQ + k] = q[k];
#pragma unroll
            for (int k = 0; k < R::NV; ++k) rowv[TP::kV + k] = v[k];
#pragma unroll
            for (int k = 0; k < R::NA; ++k) rowv[TP::kAct + k] = a[k];
            rowv[TP::kCode] = __int_as_float(TP::code(jcur, dn, jaft));
            rowv[TP::kFidx] = __int_as_float(fidx);
#pragma unroll
            for (int k = TP::kUsed; k < TP::kW; ++k) rowv[k] = 0.f;
            store_row<TP::kW>(sa.tape + ((size_t)t * p.N + env) * TP::kW, rowv);
        }
        // commit the history, then the re-initialisation (the stale pose stays, :731)
#pragma unroll
        for (int k = 0; k < 4; ++k) pose0[k] = pose[k];
        done0 = dn;
        s_ok = sm_ok;
        if (jaft >= 0) {
#pragma unroll
            for (int k = 0; k < R::NQ; ++k) q[k] = 0.f;
#pragma unroll
            for (int k = 0; k < R::NV; ++k) v[k] = 0.f;
            R::place(q, nq0, nq1);
            jcur = jaft;
            objs_ok = cfg_ok; // pool rows lie inside the placement extents
            s_ok = true;      // ... and so does the robot, at rest
        }
    }
    abuf ^= 1; // park the next block's actions
    if (l < R::NA) {
#pragma unroll
        for (int k = 0; k < kActBlock; ++k) actl[abuf][k][g][l] = anx[k];
    }
    }
    if (writer) R::store(dyn, p.Npad, env, q, v, pose0, done0, steps);
    if (live && jcur >= 0) { // the layout a reset_done installed becomes the env's layout
        float2* objw = reinterpret_cast<float2*>(obj);
        for (int o = l; o < p.nobj; o += kGL)
            objw[((size_t)(o >> 1) * p.Npad + env) * 2 + (o & 1)] = r.cand_xy[(size_t)jcur * r.nobj_total + o];
    }
}

// qpos an env starts step t+1 from, given its tape row of step t: the stepped qpos, or -- where reset_done fired --
// the rest pose at the robot position of the layout it installed (layout2qpos :623-639)
template <class R>
GX_D void next_start(const float (&rowv)[SplitTape<R>::kW], const RolloutArgs& r, float (&s)[R::NQ])
{
    using TP = SplitTape<R>;
    const int jaft = TP::jaft_of(__float_as_int(rowv[TP::kCode]));
    if (jaft >= 0 && jaft < r.n_rows) {
        const float2 rb = r.cand_xy[(size_t)jaft * r.nobj_total + r.nobj_total - 1];
#pragma unroll
        for (int k = 0; k < R::NQ; ++k) s[k] = 0.f;
        R::place(s, rb.x, rb.y);
    } else {
#pragma unroll
        for (int k = 0; k < R::NQ; ++k) s[k] = rowv[TP::kQ + k];
    }
}

// LDS tile hand-over inside a workgroup: a single wave executes its LDS operations in program order, so only the
// compiler has to keep the order (wave-scope fence over the LDS address space: no instruction, no wait for global stores)
template <int BLOCK>
GX_D void tile_sync()
{
    if (BLOCK == 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    } else {
        __syncthreads();
    }
}

template <class R, int BLOCK, int PMAX, bool kDef>
__global__ __launch_bounds__(BLOCK) void obs_tape_kernel(Params p_in, RolloutArgs r, SplitArgs sa)
{
    using TP = SplitTape<R>;
    const Params p = fold_params<R, kDef>(p_in);
    extern __shared__ float4 tile4[];
    float* tile = reinterpret_cast<float*>(tile4);
    const int tid = threadIdx.x;
    if (blockIdx.y) { // another rank's shard of the gathered buffer (one launch expands them all)
        const size_t so = (size_t)blockIdx.y * (size_t)sa.shard_stride;
        sa.tape += so; sa.entry += so;
        sa.obj0 = reinterpret_cast<float4*>(reinterpret_cast<float*>(sa.obj0) + so);
        r.obs += (size_t)blockIdx.y * (size_t)sa.out_stride;
    }
    const size_t G = (size_t)r.T * p.N;
    const int RS = r.obs_stride;
    const bool packed = r.act_out != nullptr;
    // LDS row stride: RS + 1 when RS is a multiple of 4 floats (48 for the Point's packed rows: 64 rows on 4 banks,
    // 16-way conflicts on every row write); other widths conflict 2-way at worst and keep the contiguous tile
    const int LS = (RS & 3) ? RS : RS + 1;
    float* row = tile + tid * LS;
    // GRID-STRIDE over the 64-row tiles (round 4).  With one tile per workgroup every wave of a launch goes through the same
    // three phases at about the same time -- tape loads, ~1 800 VALU instructions, a 12 KB burst of row stores; with a
    // capped grid (obs_grid) a wave's stores of tile k drain while it computes tile k + 1 (single-wave workgroups: the LDS
    // tile is reused behind a wave-scope fence, which -- unlike __syncthreads -- does not wait for the stores to be
    // acknowledged).  Measured gain at 400 000 rows: 6 % (47.1 -> 44.2 us with 3072 workgroups); a launch over several
    // shards already has that overlap between its tiles and is fastest uncapped.
    for (size_t g0 = (size_t)blockIdx.x * BLOCK; g0 < G; g0 += (size_t)gridDim.x * BLOCK) {
    const size_t g = g0 + tid;
    const bool live = g < G;
    const size_t gg = live ? g : 0;
    const int i = (int)(gg % (size_t)p.N);
    const int t = (int)(gg / (size_t)p.N);

    float rowv[TP::kW], ev[TP::kE];
    load_row<TP::kW>(sa.tape + gg * TP::kW, rowv);
    float q[R::NQ], v[R::NV], a[R::NA];
#pragma unroll
    for (int k = 0; k < R::NQ; ++k) q[k] = rowv[TP::kQ + k];
#pragma unroll
    for (int k = 0; k < R::NV; ++k) v[k] = rowv[TP::kV + k];
#pragma unroll
    for (int k = 0; k < R::NA; ++k) a[k] = rowv[TP::kAct + k];
    const int code = __float_as_int(rowv[TP::kCode]);
    const float dn = TP::done_of(code);
    const int jaft = TP::jaft_of(code);
    // the layout the step was made in: in the word itself, or -- the step finished the env -- what the rows before say
    const int jcur = code >= 0 ? code - 1 : TP::jcur_before(sa.tape, gg, (size_t)p.N, t);

    // what the step started from (s), the stale pose it found (pose0) and the done flag before it
    float s[R::NQ], pose0[4], last_done, hist0;
    if (t < 2) load_row<TP::kE>(sa.entry + (size_t)i * TP::kE, ev);
    if (t == 0) {
#pragma unroll
        for (int k = 0; k < R::NQ; ++k) s[k] = ev[TP::kEQ + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) pose0[k] = ev[TP::kEPose + k];
        last_done = ev[TP::kEDone];
    } else {
        float prev[TP::kW];
        load_row<TP::kW>(sa.tape + (gg - (size_t)p.N) * TP::kW, prev);
        next_start<R>(prev, r, s);
        last_done = TP::done_of(__float_as_int(prev[TP::kCode]));
        float sp[R::NQ]; // what the PREVIOUS step started from: its pose is this step's stale pose (reset_done keeps it, :731)
        if (t == 1) {
#pragma unroll
            for (int k = 0; k < R::NQ; ++k) sp[k] = ev[TP::kEQ + k];
        } else {
            float pp[TP::kW];
            load_row<TP::kW>(sa.tape + (gg - 2 * (size_t)p.N) * TP::kW, pp);
            next_start<R>(pp, r, sp);
        }
        R::pose_of(sp, pose0);
        world_pose(p, pose0);
    }
    hist0 = t < 2 ? ev[TP::kEHist] : 2.0f; // number of step() calls before the rollout (only its first rows care)
    const bool have_last = ((int)hist0 + t) >= 1;
    float pose[4], ctrl[R::NU];
    R::pose_of(s, pose);                  // mjx.step = forward(qpos_t); integrate: the returned xpos / xmat are one step stale
    world_pose(p, pose);
    R::convert_action(pose0, a, ctrl);    // :672-685, PRE-step xmat

    float4 ob[PMAX];
    if (jcur >= 0 && jcur < r.n_rows) { float rx_, ry_; load_layout<PMAX>(p, r.cand_xy, r.nobj_total, jcur, ob, rx_, ry_); }
    else {
#pragma unroll
        for (int k = 0; k < PMAX; ++k)
            ob[k] = (k < p.P) ? sa.obj0[(size_t)k * p.Npad + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // cost :804-811 (the layout the step was made in)
    float cs = 0.0f;
#pragma unroll
    for (int k = 0; k < PMAX; ++k) {
        if (k > 0 && 2 * k < p.nobj) cs = cs + cost_term(p, 2 * k, ob[k].x, ob[k].y, pose);
        if (2 * k + 1 < p.nobj) cs = cs + cost_term(p, 2 * k + 1, ob[k].z, ob[k].w, pose);
    }
    // reward_done :787-802 (done itself, with the NaN guard and the timeout folded in, comes from the tape)
    const float dg = dist2(ob[0].x, ob[0].y, pose[0], pose[1]);
    float last = dg;
    if (have_last && !(last_done > 0.0f)) last = dist2(ob[0].x, ob[0].y, pose0[0], pose0[1]);
    const float dd = last - dg;
    float rw = dd * p.reward_distance;
    if (fabsf(dd) > 1.0f) rw = 0.0f;
    bool bad = build_obs_row<R, PMAX>(p, row, pose, ob, ctrl, q, v, 0.f, 0.f, 0.f, 0.f);
    if (bad) rw = 0.0f;                   // :696-699
    if (jaft >= 0 && jaft < r.n_rows) { // reset_done fired: the row the learner sees is the re-initialised env's
        float rx, ry;
        load_layout<PMAX>(p, r.cand_xy, r.nobj_total, jaft, ob, rx, ry);
#pragma unroll
        for (int k = 0; k < R::NQ; ++k) q[k] = 0.f;
#pragma unroll
        for (int k = 0; k < R::NV; ++k) v[k] = 0.f;
#pragma unroll
        for (int k = 0; k < R::NU; ++k) ctrl[k] = 0.f;
        R::place(q, rx, ry);
        pose[0] = rx; pose[1] = ry; pose[2] = 1.0f; pose[3] = 0.0f;
        world_pose(p, pose);
        if constexpr (!R::kRestFixed) { // the fake step (:719-724) moves the robot: its qpos / qvel / pose feed this row
            const float* frow = r.fake + (size_t)__float_as_int(rowv[TP::kFidx]) * (R::NQ + R::NV + 4);
#pragma unroll
            for (int k = 0; k < R::NQ; ++k) q[k] = frow[k];
#pragma unroll
            for (int k = 0; k < R::NV; ++k) v[k] = frow[R::NQ + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) pose[k] = frow[R::NQ + R::NV + k];
        }
        build_obs_row<R, PMAX>(p, row, pose, ob, ctrl, q, v, 0.f, 0.f, 0.f, 0.f);
    }
    if (packed) {
#pragma unroll
        for (int k = 0; k < R::NA; ++k) row[p.D + k] = a[k];
        row[p.D + R::NA] = rw; row[p.D + R::NA + 1] = cs; row[p.D + R::NA + 2] = dn;
    } else if (live) {
        r.rew[g] = rw; r.cost[g] = cs; r.done[g] = dn;
    }
    tile_sync<BLOCK>();
    const size_t left = G - g0;
    const int nrow = left < (size_t)BLOCK ? (int)left : BLOCK;
    if (LS == RS) flush_tile<BLOCK>(tile, r.obs + g0 * RS, nrow * RS);
    else if ((reinterpret_cast<uintptr_t>(r.obs) & 15u) == 0) flush_tile_padded<BLOCK>(tile, LS, r.obs + g0 * RS, nrow, RS);
    else {
        for (int k = tid; k < nrow * RS; k += BLOCK) { const int rw_ = k / RS; r.obs[g0 * RS + k] = tile[rw_ * LS + (k - rw_ * RS)]; }
    }
    tile_sync<BLOCK>(); // the tile is rewritten by the next iteration
    }
}

// workgroups of the observation pass per shard.  ALONE on the chip: enough to fill it a few waves deep, few enough that
// every wave takes several tiles in turn (kObsGridCap; the 6 % of the comment in obs_tape_kernel).  With COMPANY -- a
// layout sampler of this engine in flight beside the rollout, or another rank's work -- one workgroup per tile: the
// pass then competes with throughput kernels whose waves fill every SIMD, and its share of the issue slots is its share
// of the resident waves.  Same-box A/B in the epoch (tools/ab_epoch.py, round 5): Swimmer, whose observation pass is on
// the epoch's critical chain, 688 -> 720 M env-steps/s uncapped (this cap, introduced in round 4 after a standalone
// A/B, was the Swimmer's round-3 -> round-4 regression, 706 -> 689 M on one box); Point 773 = 773, Ant 300 = 300.
// (GX_OBS_GRID_CAP: experiments.)
// (robots opt out of the uncapped form with `static constexpr bool kObsCapBesideSampler = true`: the Point)
template <class R, class = void> struct ObsCapTrait { static constexpr bool value = false; };
template <class R> struct ObsCapTrait<R, std::void_t<decltype(R::kObsCapBesideSampler)>> { static constexpr bool value = R::kObsCapBesideSampler; };
template <class R> constexpr bool obs_cap_beside_sampler() { return ObsCapTrait<R>::value; }
static unsigned obs_grid(size_t rows, int block, int n_shards, bool company)
{
    static const int forced = [] { const char* e = getenv("GX_OBS_GRID_CAP"); return e ? atoi(e) : 0; }();
    const size_t tiles = (rows + block - 1) / block;
    if (forced <= 0 && (n_shards > 1 || company)) return (unsigned)tiles;
    size_t cap = forced > 0 ? (size_t)forced : (size_t)kObsGridCap;
    cap = (cap + n_shards - 1) / n_shards;
    if (cap < 1) cap = 1;
    return (unsigned)(tiles < cap ? tiles : cap);
}

// which: bit 0 = the dynamics pass, bit 1 = the observation pass (gx_rollout: both; the tape hand-off runs them on
// different ranks: gx_rollout_tape / gx_expand_tape)
template <class R, int PMAX>
static hipError_t launch_split_p(const Params& p, const RolloutArgs& r, const SplitArgs& sa, const DevBuffers& b, hipStream_t s,
                                hipEvent_t hold, int which, int n_shards = 1)
{
    hipError_t st = hipSuccess; // of the wait that orders the observation pass behind the sampler: must not be dropped
    constexpr int B1 = 64, B2 = 64;
    const int lpe = (R::kDynLanes == 4 && sa.lanes == 4) ? 4 : 1;
    const dim3 g1((p.N * lpe + B1 - 1) / B1), g2(obs_grid((size_t)r.T * p.N, B2, n_shards, sa.lanes != 4 && !obs_cap_beside_sampler<R>()), n_shards);
    const size_t lds1 = sizeof(float) * ((size_t)B1 * p.D + 2 * (size_t)kActBlock * B1 * R::NA); // obs rows + two action blocks
    const size_t lds2 = sizeof(float) * (size_t)B2 * (r.obs_stride | 1);
    const bool def = PMAX == 5 && is_default_layout<R>(p);
    if (which & 1) {
        if constexpr (R::kDynLanes == 4) {
            if (lpe == 4) {
                if (def) hipLaunchKernelGGL((dyn_tape_kernel<R, B1, 5, true, 4>), g1, dim3(B1), lds1, s, p, r, sa, b.dyn, b.obj);
                else hipLaunchKernelGGL((dyn_tape_kernel<R, B1, PMAX, false, 4>), g1, dim3(B1), lds1, s, p, r, sa, b.dyn, b.obj);
            }
        }
        if (lpe == 1) {
            if (def) hipLaunchKernelGGL((dyn_tape_kernel<R, B1, 5, true, 1>), g1, dim3(B1), lds1, s, p, r, sa, b.dyn, b.obj);
            else hipLaunchKernelGGL((dyn_tape_kernel<R, B1, PMAX, false, 1>), g1, dim3(B1), lds1, s, p, r, sa, b.dyn, b.obj);
        }
    }
    if (hold) st = hipStreamWaitEvent(s, hold, 0);
    if (st != hipSuccess) return st;
    if (which & 2) {
        if (def) hipLaunchKernelGGL((obs_tape_kernel<R, B2, 5, true>), g2, dim3(B2), lds2, s, p, r, sa);
        else hipLaunchKernelGGL((obs_tape_kernel<R, B2, PMAX, false>), g2, dim3(B2), lds2, s, p, r, sa);
    }
    return st;
}

// the same for the robots whose dynamics pass is the lane-group kernel (Ant, Walker)
template <class R, int PMAX>
static hipError_t launch_split_group_p(const Params& p, const RolloutArgs& r, const SplitArgs& sa, const DevBuffers& b,
                                       hipStream_t s, hipEvent_t hold, int which, int n_shards = 1)
{
    hipError_t st = hipSuccess;
    constexpr int B2 = 64;
    // Envs per wave: as few as still leave one wave per SIMD (1024 on the chip) -- one up to 1024 envs, two up to 2048,
    // four beyond.  A wave runs the step's data-dependent loops (the active-set Newton iterations above all) until its
    // LAST env is done, and it issues nearly every slot it gets (two waves on one SIMD take 1.97x the time of one), so
    // fewer envs per wave is faster exactly as long as the waves do not have to share SIMDs.  Ant, 200 steps, round 4:
    // 2000 envs 1262 us (4 per wave) -> 1205 us (2); 1000 envs 1131 us (1); Walker 2361 -> 2279 us.
    // (GX_GROUP_EPW = 1 | 2 | 4: experiments.)
    // ... and only ALONE on the chip (sa.lanes == 4: no layout sampler of this engine in flight, as for the Swimmer's
    // quad form): beside the sampler the four-env waves leave it half of the SIMDs to itself -- Ant epoch 1.53 -> 1.46 ms
    // (262 -> 274 M env-steps/s), the 18-object config 5 2.79 -> 2.67 ms with four.
    static const int epw_forced = [] { const char* e = getenv("GX_GROUP_EPW"); return e ? atoi(e) : 0; }();
    const int epw = epw_forced == 1 || epw_forced == 2 || epw_forced == 4
                        ? epw_forced
                        : (sa.lanes != 4 ? 4 : (p.N <= 1024 ? 1 : ((p.N + 1) / 2 <= 1024 ? 2 : 4)));
    SplitArgs sg = sa;
    sg.lanes = epw;
    const dim3 g1((p.N + epw - 1) / epw), g2(obs_grid((size_t)r.T * p.N, B2, n_shards, sa.lanes != 4), n_shards);
    const size_t lds2 = sizeof(float) * (size_t)B2 * (r.obs_stride | 1);
    if (which & 1) {
#define GX_GDYN_LAUNCH(OPL, BPL, DEF) \
    hipLaunchKernelGGL((group_dyn_tape_kernel<R, OPL, BPL, DEF>), g1, dim3(64), 0, s, p, r, sg, b.dyn, b.obj)
        if (is_default_layout<R>(p)) GX_GDYN_LAUNCH(1, 1, true);
        else if (p.nobj <= 16 && p.bins <= 16) GX_GDYN_LAUNCH(1, 1, false);
        else if (p.nobj <= 32 && p.bins <= 16) GX_GDYN_LAUNCH(2, 1, false);
        else GX_GDYN_LAUNCH(5, 4, false);
#undef GX_GDYN_LAUNCH
    }
    if (hold) st = hipStreamWaitEvent(s, hold, 0);
    if (st != hipSuccess) return st;
    if (which & 2) {
        if (PMAX == 5 && is_default_layout<R>(p)) hipLaunchKernelGGL((obs_tape_kernel<R, B2, 5, true>), g2, dim3(B2), lds2, s, p, r, sa);
        else hipLaunchKernelGGL((obs_tape_kernel<R, B2, PMAX, false>), g2, dim3(B2), lds2, s, p, r, sa);
    }
    return st;
}

} // namespace gx
