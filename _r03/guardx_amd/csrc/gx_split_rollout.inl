// gx_split_rollout.inl -- the fused T-step rollout at small env_num as TWO kernels instead of one persistent lane-group
// kernel (included by gx_robot_kernels.inl):
//
//   pass 1  the serial chain: convert_action, mjx.step, done / NaN guard / timeout, reset_done (layout index draw +
//           re-placement) -- everything the NEXT step depends on -- and one SLIM tape row per (step, env): qpos, qvel after
//           the step, the action, done, the layout row in effect and the layout row a reset_done installed.
//           dyn_tape_kernel (Point, Swimmer): one thread per env, 12 / 16 floats per row, ~170 instructions per Point step
//           (the Swimmer also as a quad of lanes per env, SwimmerRobot::substep_q).
//           group_dyn_tape_kernel (Ant, Walker; round 3): the lane-group form of their step, 16 lanes per env, 36 / 40
//           floats per row (+ the row of the pool's fake-step table a reset_done observation is read from).
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

namespace gx {

template <class R>
struct SplitTape {
    // kFidx (Ant, Walker): row of Pool::fake (= index into the compacted layout list) of the layout a reset_done installed
    static constexpr int kQ = 0, kV = kQ + R::NQ, kAct = kV + R::NV, kDone = kAct + R::NA, kJcur = kDone + 1,
                         kJaft = kJcur + 1, kFidx = kJaft + 1, kUsed = kFidx + (R::kRestFixed ? 0 : 1),
                         kW = (kUsed + 3) / 4 * 4;
    // entry record of an env: qpos at entry | the stale pose (x, y, cos, sin) | done0 | number of step() calls so far
    static constexpr int kEQ = 0, kEPose = R::NQ, kEDone = kEPose + 4, kEHist = kEDone + 1, kE = (kEHist + 1 + 3) / 4 * 4;
    static_assert(!R::kRestFixed || kE == 12, "entry record of the light robots: 12 floats (include/guardx.h)");
};

struct SplitArgs {
    float* tape;        // [T][N][kW]
    float4* obj0;       // [P][Npad] snapshot of the layouts at entry (pass 2 reads it for rows with jcur < 0)
    float* entry;       // [N][kE] state at entry (pass 2 needs it for the rows of steps 0 and 1)
    int lanes;          // lanes per env in pass 1 where the robot offers a choice (R::kDynLanes): 1 or 4
};

GX_D bool moderate(float x) { return fabsf(x) < 1e18f; } // false for NaN / Inf too
constexpr int kActBlock = 16; // steps whose actions the dynamics pass fetches at once

template <int W>
GX_D void load_row(const float* __restrict__ p, float (&v)[W])
{
#pragma unroll
    for (int k = 0; k < W / 4; ++k) {
        const float4 t4 = reinterpret_cast<const float4*>(p)[k];
        v[4 * k] = t4.x; v[4 * k + 1] = t4.y; v[4 * k + 2] = t4.z; v[4 * k + 3] = t4.w;
    }
}
template <int W>
GX_D void store_row(float* __restrict__ p, const float (&v)[W])
{
#pragma unroll
    for (int k = 0; k < W / 4; ++k)
        reinterpret_cast<float4*>(p)[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
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
    bool s_ok, p_ok;
    {
        float m0 = fabsf(pose0[2]) + fabsf(pose0[3]), m1 = 0.f;
#pragma unroll
        for (int k = 0; k < R::NQ; ++k) m1 = m1 + fabsf(q[k]);
#pragma unroll
        for (int k = 0; k < R::NV; ++k) m1 = m1 + fabsf(v[k]);
        s_ok = moderate(m1); p_ok = moderate(m0);
    }
#pragma unroll 1
    for (int tb = 0; tb < r.T; tb += kActBlock) { // blocks of kActBlock steps (two loops: the block's addresses are
                                                  // computed once per block, not carried through every step)
    if (tb + kActBlock < r.T) { // request the next block's actions (consumed kActBlock steps from now)
#pragma unroll
        for (int k = 0; k < kActBlock; ++k)
            if (tb + kActBlock + k < r.T) load_action<R>(r.act, (size_t)(tb + kActBlock + k) * p.N + i, anx[k]);
    }
    const int kend = r.T - tb < kActBlock ? r.T - tb : kActBlock;
#pragma unroll 1
    for (int kb = 0; kb < kend; ++kb) {
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
        float mag = 0.f;
#pragma unroll
        for (int k = 0; k < R::NQ; ++k) mag = mag + fabsf(qf[k]);
#pragma unroll
        for (int k = 0; k < R::NV; ++k) mag = mag + fabsf(vf[k]);
#pragma unroll
        for (int k = 0; k < R::NA; ++k) mag = mag + fabsf(a[k]);
        const float mx = pose[0] - pose0[0], my = pose[1] - pose0[1];
        const float gdx = gx - pose[0], gdy = gy - pose[1];
        const float d2 = gdx * gdx + gdy * gdy;                 // dist2()'s radicand
        const bool ordinary = objs_ok && s_ok && p_ok && p.physics_steps == 1 && moderate(mag) &&
                              (mx * mx + my * my) < 0.9f && d2 < 1e8f;
        float dn;
        if (ordinary) {
#pragma unroll
            for (int k = 0; k < R::NQ; ++k) q[k] = qf[k];
#pragma unroll
            for (int k = 0; k < R::NV; ++k) v[k] = vf[k];
            dn = d2 < p.goal_cut ? 1.0f : 0.0f;
            p_ok = true;   // this step's pose: the kinematics of a moderate qpos
            s_ok = true;   // moderate(mag)
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
            p_ok = moderate(fabsf(pose[2]) + fabsf(pose[3]));
            s_ok = moderate(m1);
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
        }

        // tape row
        float rowv[TP::kW];
#pragma unroll
        for (int k = 0; k < R::NQ; ++k) rowv[TP::kQ + k] = q[k];
#pragma unroll
        for (int k = 0; k < R::NV; ++k) rowv[TP::kV + k] = v[k];
#pragma unroll
        for (int k = 0; k < R::NA; ++k) rowv[TP::kAct + k] = a[k];
        rowv[TP::kDone] = dn;
        rowv[TP::kJcur] = __int_as_float(jcur); rowv[TP::kJaft] = __int_as_float(jaft);
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
            s_ok = true;      // ... and so does the robot, at rest
        }
    }
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
    const int env = blockIdx.x * EPW + g;
    const bool live = env < p.N;
    const int e = live ? env : 0;
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
        if (r.do_reset && live && dn > 0.0f && L > 0) {
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
            rowv[TP::kDone] = dn;
            rowv[TP::kJcur] = __int_as_float(jcur); rowv[TP::kJaft] = __int_as_float(jaft);
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
    const int jaft = __float_as_int(rowv[TP::kJaft]);
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

template <class R, int BLOCK, int PMAX, bool kDef>
__global__ __launch_bounds__(BLOCK) void obs_tape_kernel(Params p_in, RolloutArgs r, SplitArgs sa)
{
    using TP = SplitTape<R>;
    const Params p = fold_params<R, kDef>(p_in);
    extern __shared__ float4 tile4[];
    float* tile = reinterpret_cast<float*>(tile4);
    const int tid = threadIdx.x;
    const size_t G = (size_t)r.T * p.N;
    const size_t g0 = (size_t)blockIdx.x * BLOCK, g = g0 + tid;
    const bool live = g < G;
    const size_t gg = live ? g : 0;
    const int i = (int)(gg % (size_t)p.N);
    const int t = (int)(gg / (size_t)p.N);
    const int RS = r.obs_stride;
    const bool packed = r.act_out != nullptr;
    // LDS row stride: RS + 1 when RS is a multiple of 4 floats (48 for the Point's packed rows: 64 rows on 4 banks,
    // 16-way conflicts on every row write); other widths conflict 2-way at worst and keep the contiguous tile
    const int LS = (RS & 3) ? RS : RS + 1;
    float* row = tile + tid * LS;

    float rowv[TP::kW], ev[TP::kE];
    load_row<TP::kW>(sa.tape + gg * TP::kW, rowv);
    float q[R::NQ], v[R::NV], a[R::NA];
#pragma unroll
    for (int k = 0; k < R::NQ; ++k) q[k] = rowv[TP::kQ + k];
#pragma unroll
    for (int k = 0; k < R::NV; ++k) v[k] = rowv[TP::kV + k];
#pragma unroll
    for (int k = 0; k < R::NA; ++k) a[k] = rowv[TP::kAct + k];
    const float dn = rowv[TP::kDone];
    const int jcur = __float_as_int(rowv[TP::kJcur]), jaft = __float_as_int(rowv[TP::kJaft]);

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
        last_done = prev[TP::kDone];
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
    __syncthreads();
    const size_t left = G - g0;
    const int nrow = left < (size_t)BLOCK ? (int)left : BLOCK;
    if (LS == RS) flush_tile<BLOCK>(tile, r.obs + g0 * RS, nrow * RS);
    else if ((reinterpret_cast<uintptr_t>(r.obs) & 15u) == 0) flush_tile_padded<BLOCK>(tile, LS, r.obs + g0 * RS, nrow, RS);
    else {
        for (int k = tid; k < nrow * RS; k += BLOCK) { const int rw_ = k / RS; r.obs[g0 * RS + k] = tile[rw_ * LS + (k - rw_ * RS)]; }
    }
}

// which: bit 0 = the dynamics pass, bit 1 = the observation pass (gx_rollout: both; the tape hand-off runs them on
// different ranks: gx_rollout_tape / gx_expand_tape)
template <class R, int PMAX>
static hipError_t launch_split_p(const Params& p, const RolloutArgs& r, const SplitArgs& sa, const DevBuffers& b, hipStream_t s,
                                hipEvent_t hold, int which)
{
    hipError_t st = hipSuccess; // of the wait that orders the observation pass behind the sampler: must not be dropped
    constexpr int B1 = 64, B2 = 64;
    const int lpe = (R::kDynLanes == 4 && sa.lanes == 4) ? 4 : 1;
    const dim3 g1((p.N * lpe + B1 - 1) / B1), g2((unsigned)(((size_t)r.T * p.N + B2 - 1) / B2));
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
                                       hipStream_t s, hipEvent_t hold, int which)
{
    hipError_t st = hipSuccess;
    constexpr int B2 = 64;
    const dim3 g1((p.N + 3) / 4), g2((unsigned)(((size_t)r.T * p.N + B2 - 1) / B2));
    const size_t lds2 = sizeof(float) * (size_t)B2 * (r.obs_stride | 1);
    if (which & 1) {
#define GX_GDYN_LAUNCH(OPL, BPL, DEF) \
    hipLaunchKernelGGL((group_dyn_tape_kernel<R, OPL, BPL, DEF>), g1, dim3(64), 0, s, p, r, sa, b.dyn, b.obj)
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
