// gx_robot_kernels.inl -- the robot-templated kernels of the GUARD batched environment step (gfx950)
// and their launchers.  Included by one translation unit per robot (gx_kernels_<robot>.hip), each of
// which instantiates RobotLaunch<R> for its robot, so the robots compile in parallel.
//
// One thread owns one environment.  Environment state is struct-of-arrays of
// float4 (`dyn`, `obj`), so each wave64 load/store is one coalesced 1 KiB
// transaction.  The learner-facing observation is env-major (N, D) (the learner
// writes obs_buf[:, t, :] = obs, reference trpo.py:58), so every thread builds
// its D-float row in LDS -- the lidar bins are scatter-max'ed in place there --
// and the block then streams the whole tile out as contiguous float4 stores.
//
// Reference lines: /root/reference/safe_rl_envs/safe_rl_envs/envs/engine.py.
#pragma once
#include "gx_kernels.h"
#include "gx_robot.h"
#include "gx_policy.h"

namespace gx {

// ---------------------------------------------------------------------------
// observation row (engine.py:738-778) built in this thread's LDS row.
// `ob` holds the object pairs: ob[k] = (obj 2k xy, obj 2k+1 xy); obj 0 = goal.
// ---------------------------------------------------------------------------
template <class R, int PMAX>
GX_D bool build_obs_row(const Params& p, float* row, const float (&pose)[4],
                        const float4 (&ob)[PMAX], const float (&ctrl)[R::NU], const float (&q)[R::NQ],
                        const float (&v)[R::NV], float vel0, float vel1, float acc0, float acc1)
{
    bool bad = false;
    if (p.off_acc >= 0) {
        row[p.off_acc] = acc0; row[p.off_acc + 1] = acc1;
        bad = bad || notfinite(acc0) || notfinite(acc1);
    }
    if (p.off_ctrl >= 0) {
#pragma unroll
        for (int k = 0; k < R::NU; ++k) { row[p.off_ctrl + k] = ctrl[k]; bad = bad || notfinite(ctrl[k]); }
    }
    if (p.off_comp >= 0) { // obs_compass :834-844
        const float dx = ob[0].x - pose[0], dy = ob[0].y - pose[1];
        const float zx = dx * pose[2] + dy * pose[3];
        const float zy = dx * (-pose[3]) + dy * pose[2];
        row[p.off_comp] = zx; row[p.off_comp + 1] = zy;
        bad = bad || notfinite(zx) || notfinite(zy);
    }
    if (p.off_gl >= 0) {
        float* r = row + p.off_gl;
        for (int b = 0; b < p.bins; ++b) r[b] = 0.0f;
        bad = lidar_one(p, r, ob[0].x, ob[0].y, pose) || bad;
    }
    if (p.off_hl >= 0 || p.off_pl >= 0) {
        float* rh = row + (p.off_hl >= 0 ? p.off_hl : 0);
        float* rp = row + (p.off_pl >= 0 ? p.off_pl : 0);
        if (p.off_hl >= 0) for (int b = 0; b < p.bins; ++b) rh[b] = 0.0f;
        if (p.off_pl >= 0) for (int b = 0; b < p.bins; ++b) rp[b] = 0.0f;
#pragma unroll
        for (int k = 0; k < PMAX; ++k) {
            // objects 2k and 2k+1; object 0 is the goal, 1..H hazards, H+1.. pillars
            const int oa = 2 * k, ob_ = 2 * k + 1;
            if (k > 0 && oa < p.nobj) {
                if (oa <= p.H) { if (p.off_hl >= 0) bad = lidar_one(p, rh, ob[k].x, ob[k].y, pose) || bad; }
                else if (p.off_pl >= 0) bad = lidar_one(p, rp, ob[k].x, ob[k].y, pose) || bad;
            }
            if (ob_ < p.nobj) {
                if (ob_ <= p.H) { if (p.off_hl >= 0) bad = lidar_one(p, rh, ob[k].z, ob[k].w, pose) || bad; }
                else if (p.off_pl >= 0) bad = lidar_one(p, rp, ob[k].z, ob[k].w, pose) || bad;
            }
        }
    }
    if (p.off_qpos >= 0) {
#pragma unroll
        for (int k = 0; k < R::NQ; ++k) { row[p.off_qpos + k] = q[k]; bad = bad || notfinite(q[k]); }
    }
    if (p.off_qvel >= 0) {
#pragma unroll
        for (int k = 0; k < R::NV; ++k) { row[p.off_qvel + k] = v[k]; bad = bad || notfinite(v[k]); }
    }
    if (p.off_vel >= 0) {
        row[p.off_vel] = vel0; row[p.off_vel + 1] = vel1;
        bad = bad || notfinite(vel0) || notfinite(vel1);
    }
    return bad;
}

// stream the block's LDS tile (nenv rows of D floats, env-major) to global.
// float4 stores when the destination is 16-byte aligned (always true for an
// (N, D) tensor; a time-major slice with odd N*D may not be), dwords otherwise.
template <int BLOCK>
GX_D void flush_tile(const float* tile, float* gbase, int total)
{
    if ((reinterpret_cast<uintptr_t>(gbase) & 15u) == 0) {
        const int nvec = total >> 2;
        const float4* t4 = reinterpret_cast<const float4*>(tile);
        float4* g4 = reinterpret_cast<float4*>(gbase);
        for (int v = threadIdx.x; v < nvec; v += BLOCK) g4[v] = t4[v];
        for (int k = (nvec << 2) + threadIdx.x; k < total; k += BLOCK) gbase[k] = tile[k];
    } else {
        for (int k = threadIdx.x; k < total; k += BLOCK) gbase[k] = tile[k];
    }
}

// The same for a tile whose rows sit `ls` floats apart in LDS while the destination rows are `rs` floats (rs % 4 == 0,
// 16-byte aligned destination): a row stride that is a multiple of 16 floats (the 48-float packed hand-off row) puts
// the 64 rows of a wave on 4 LDS banks, 16-way conflicts on every row write; ls = rs + 1 spreads them.
template <int BLOCK>
GX_D void flush_tile_padded(const float* tile, int ls, float* gbase, int nrow, int rs)
{
    const int q4 = rs >> 2, nvec = nrow * q4;
    const unsigned inv = (1u << 20) / (unsigned)q4 + 1u; // v / q4 for v < 2^20 / q4 ... exact here (v < 64 * 64)
    float4* g4 = reinterpret_cast<float4*>(gbase);
    for (int v = threadIdx.x; v < nvec; v += BLOCK) {
        const int row = (int)(((unsigned)v * inv) >> 20), c = (v - row * q4) << 2;
        const float* t = tile + row * ls + c;
        g4[v] = make_float4(t[0], t[1], t[2], t[3]);
    }
}

template <int BLOCK>
GX_D void stage_tile(float* tile, const float* gbase, int total)
{
    if ((reinterpret_cast<uintptr_t>(gbase) & 15u) == 0) {
        const int nvec = total >> 2;
        float4* t4 = reinterpret_cast<float4*>(tile);
        const float4* g4 = reinterpret_cast<const float4*>(gbase);
        for (int v = threadIdx.x; v < nvec; v += BLOCK) t4[v] = g4[v];
        for (int k = (nvec << 2) + threadIdx.x; k < total; k += BLOCK) tile[k] = gbase[k];
    } else {
        for (int k = threadIdx.x; k < total; k += BLOCK) tile[k] = gbase[k];
    }
}

GX_D float dist2(float ax, float ay, float bx, float by)
{
    const float dx = ax - bx, dy = ay - by;
    return sqrtf(dx * dx + dy * dy);
}

// Fold the integer layout of the default Goal_<Robot>_8Hazards observation (8 hazards, 16
// bins, every observe_* flag at its default, aliasing on, exponential lidar, 1 physics step)
// into compile-time constants: loops unroll, flag tests and their scalar bookkeeping disappear.
template <class R>
static bool is_default_layout(const Params& p)
{
    return p.nobj == 9 && p.PL == 0 && p.off_pl == -1 && p.bins == 16 && p.D == R::kD && p.off_acc == -1 && p.off_ctrl == R::kOffCtrl &&
           p.off_comp == R::kOffComp && p.off_gl == R::kOffGl && p.off_hl == R::kOffHl &&
           p.off_qpos == R::kOffQpos && p.off_qvel == R::kOffQvel && p.off_vel == -1 && p.lidar_alias == 1 &&
           p.lidar_max_dist_set == 0 && p.physics_steps == 1 && p.hist_on == 0 && p.rot_on == 0;
}

template <class R, bool kDef>
GX_D Params fold_params(Params p)
{
    if (kDef) {
        p.H = 8; p.PL = 0; p.off_pl = -1; p.nobj = 9; p.P = 5; p.bins = 16; p.D = R::kD;
        p.off_acc = -1; p.off_ctrl = R::kOffCtrl; p.off_comp = R::kOffComp; p.off_gl = R::kOffGl;
        p.off_hl = R::kOffHl; p.off_qpos = R::kOffQpos; p.off_qvel = R::kOffQvel; p.off_vel = -1;
        p.lidar_alias = 1; p.lidar_max_dist_set = 0; p.physics_steps = 1; p.hist_on = 0; p.rot_on = 0;
    }
    return p;
}

// ego_vel_acc (engine.py:902-929)
GX_D void ego_vel_acc(const Params& p, const float (&pose)[4], float L1x, float L1y, float P2x, float P2y,
                      float last_done, float done2, bool have_last, bool have_last_last, float& vel0,
                      float& vel1, float& acc0, float& acc1)
{
    float plx = pose[0], ply = pose[1], pllx = pose[0], plly = pose[1];
    if (have_last) {
        if (!(last_done > 0.0f)) { plx = L1x; ply = L1y; }
        if (have_last_last) {
            if (done2 + last_done > 0.0f) { pllx = plx; plly = ply; }
            else { pllx = P2x; plly = P2y; }
        }
    }
    const float vwx = (pose[0] - plx) / p.dt, vwy = (pose[1] - ply) / p.dt;
    const float lvx = (plx - pllx) / p.dt, lvy = (ply - plly) / p.dt;
    const float awx = (vwx - lvx) / p.dt, awy = (vwy - lvy) / p.dt;
    vel0 = vwx * pose[2] + vwy * pose[3];
    vel1 = vwx * (-pose[3]) + vwy * pose[2];
    acc0 = awx * pose[2] + awy * pose[3];
    acc1 = awx * (-pose[3]) + awy * pose[2];
}

// action row of env `i` ((N, NA) row-major): one 8/16-byte load per 2/4 floats when the base allows it
template <class R>
GX_D void load_action(const float* __restrict__ act, size_t i, float (&a)[R::NA])
{
    const float* row = act + i * R::NA;
    if (R::NA == 2) {
        const float2 t = *reinterpret_cast<const float2*>(row); // every entry point checks 8-byte alignment
        a[0] = t.x; a[1] = t.y;
    } else if (R::NA % 4 == 0 && (reinterpret_cast<uintptr_t>(act) & 15u) == 0) {
#pragma unroll
        for (int k = 0; k < R::NA / 4; ++k) {
            const float4 t = reinterpret_cast<const float4*>(row)[k];
            a[4 * k] = t.x; a[4 * k + 1] = t.y; a[4 * k + 2] = t.z; a[4 * k + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < R::NA; ++k) a[k] = row[k];
    }
}

// ---------------------------------------------------------------------------
// Engine.step (engine.py:469-495 + mjx_step :659-700), thread-per-env form.
// ---------------------------------------------------------------------------
template <class R, int BLOCK, int PMAX, bool kQacc, bool kDef>
__global__ __launch_bounds__(BLOCK) void step_kernel(Params p_in, const float* __restrict__ act,
                                                     float4* __restrict__ dyn,
                                                     const float4* __restrict__ obj,
                                                     float4* __restrict__ hist,
                                                     float* __restrict__ obs,
                                                     float* __restrict__ rew,
                                                     float* __restrict__ cost,
                                                     float* __restrict__ done,
                                                     float* __restrict__ qacc_out)
{
    const Params p = fold_params<R, kDef>(p_in);
    extern __shared__ float4 tile4[];
    float* tile = reinterpret_cast<float*>(tile4);
    const int tid = threadIdx.x;
    const int env0 = blockIdx.x * BLOCK;
    const int i = env0 + tid;
    const bool live = i < p.N;

    // ---- coalesced loads (arrays are padded to Npad, every lane may load)
    float a[R::NA];
    load_action<R>(act, live ? i : 0, a);
    float q[R::NQ], v[R::NV], pose0[4], last_done, steps;
    R::load(dyn, p.Npad, i, q, v, pose0, last_done, steps);
    float4 ob[PMAX];
#pragma unroll
    for (int k = 0; k < PMAX; ++k)
        ob[k] = (k < p.P) ? obj[(size_t)k * p.Npad + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 hs = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.hist_on) hs = hist[i];
    const float P1x = pose0[0], P1y = pose0[1]; // last_data.xpos

    float ctrl[R::NU];
    R::convert_action(pose0, a, ctrl); // :672-685, PRE-step xmat
    float pose[4], qacc[R::NV];
#pragma unroll
    for (int k = 0; k < R::NV; ++k) qacc[k] = 0.f;
    for (int k = 0; k < p.physics_steps; ++k) R::template substep<kQacc>(q, v, ctrl, pose, qacc);
    world_pose(p, pose);

    float vel0 = 0.f, vel1 = 0.f, acc0 = 0.f, acc1 = 0.f;
    if (p.hist_on)
        ego_vel_acc(p, pose, P1x, P1y, hs.x, hs.y, last_done, hs.z, p.have_last != 0, p.have_last_last != 0,
                    vel0, vel1, acc0, acc1);

    float* row = tile + tid * p.D;
    const bool bad = build_obs_row<R, PMAX>(p, row, pose, ob, ctrl, q, v, vel0, vel1, acc0, acc1);

    // reward_done :787-802
    const float dg = dist2(ob[0].x, ob[0].y, pose[0], pose[1]);
    float last = dg;
    if (p.have_last && !(last_done > 0.0f)) last = dist2(ob[0].x, ob[0].y, P1x, P1y);
    const float dd = last - dg;
    float r = dd * p.reward_distance;
    float dn = dg < p.goal_size ? 1.0f : 0.0f;
    if (fabsf(dd) > 1.0f) { dn = 1.0f; r = 0.0f; }

    // cost :804-811
    float cs = 0.0f;
#pragma unroll
    for (int k = 0; k < PMAX; ++k) {
        if (k > 0 && 2 * k < p.nobj) cs = cs + cost_term(p, 2 * k, ob[k].x, ob[k].y, pose);
        if (2 * k + 1 < p.nobj) cs = cs + cost_term(p, 2 * k + 1, ob[k].z, ob[k].w, pose);
    }

    // NaN/Inf guard :696-699, timeout + step counter :492-493
    if (bad) { r = 0.0f; dn = 1.0f; }
    if (steps > p.num_steps_f) dn = 1.0f;
    const float nsteps = dn > 0.0f ? 0.0f : steps + 1.0f;

    if (live) {
        R::store(dyn, p.Npad, i, q, v, pose, dn, nsteps);
        if (p.hist_on) hist[i] = make_float4(P1x, P1y, last_done, 0.f);
        rew[i] = r;
        cost[i] = cs;
        done[i] = dn;
        if (kQacc) {
#pragma unroll
            for (int k = 0; k < R::NV; ++k) qacc_out[R::NV * i + k] = qacc[k];
        }
    }

    __syncthreads();
    const int nenv = min(BLOCK, p.N - env0);
    flush_tile<BLOCK>(tile, obs + (size_t)env0 * p.D, nenv * p.D);
}

// synchronise the workgroup's LDS traffic only (one-wave workgroups: the wave's LDS operations execute in program
// order, a compiler-only fence; larger ones: s_waitcnt lgkmcnt(0) + s_barrier) -- no wait for outstanding global stores
template <int BLOCK>
GX_D void lds_sync()
{
    if (BLOCK == 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    } else {
        wg_sync_lds();
    }
}

// ---------------------------------------------------------------------------
// Engine.reset (engine.py:454-467): get_layout + mjx_reset for every env
// ---------------------------------------------------------------------------
template <int PMAX>
GX_D void load_layout(const Params& p, const float2* __restrict__ cand_xy, int nobj_total, int j,
                      float4 (&ob)[PMAX], float& rx, float& ry)
{
    const float2* rowp = cand_xy + (size_t)j * nobj_total;
#pragma unroll
    for (int k = 0; k < PMAX; ++k) {
        float2 a = make_float2(0.f, 0.f), b = make_float2(0.f, 0.f);
        if (2 * k < p.nobj) a = rowp[2 * k];
        if (2 * k + 1 < p.nobj) b = rowp[2 * k + 1];
        ob[k] = make_float4(a.x, a.y, b.x, b.y);
    }
    const float2 rb = rowp[nobj_total - 1];
    rx = rb.x; ry = rb.y;
}

template <class R, int BLOCK, int PMAX>
__global__ __launch_bounds__(BLOCK) void reset_apply_kernel(Params p, int nobj_total, uint32_t k10,
                                                            uint32_t k11, uint32_t k20, uint32_t k21,
                                                            const int* __restrict__ layout_size,
                                                            const int* __restrict__ cand_of,
                                                            const float2* __restrict__ cand_xy,
                                                            float4* __restrict__ dyn,
                                                            float4* __restrict__ obj,
                                                            float* __restrict__ obs,
                                                            int* __restrict__ host_layout_size)
{
    extern __shared__ float4 tile4[];
    float* tile = reinterpret_cast<float*>(tile4);
    // One wave per SIMD on 32 SIMDs, in the serial chain of the epoch (dyn(k) -> reset_apply(k + 1) -> dyn(k + 1)) and
    // always beside the prefetch sampler's phase 2, whose waves fill every SIMD: without priority its ~2 k dependent
    // instructions wait their turn in the oldest-first arbitration (37 us for the Point, 78 us for the Swimmer in the
    // epoch's kernel trace against ~8 us alone).
    __builtin_amdgcn_s_setprio(3);
    const int L = *layout_size;
    // len(idx) for the host-side assert (engine.py:442-444): written straight into mapped pinned
    // host memory, no copy kernel on the stream
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        host_layout_size[0] = L;
        if (L < host_layout_size[1]) host_layout_size[1] = L; // smallest pool since the last check (resets are stream ordered)
    }
    if (L <= 0) return; // host raises GX_ERR_LAYOUT (engine.py:444)
    const int tid = threadIdx.x;
    const int env0 = blockIdx.x * BLOCK;
    const int i = env0 + tid;
    const bool live = i < p.N;
    const uint32_t gi = (uint32_t)(p.env_offset + (live ? i : 0));
    const uint32_t idx = randint_at(k10, k11, k20, k21, (uint32_t)p.env_total, (uint32_t)L, gi);
    const int j = cand_of[idx];
    float4 ob[PMAX];
    float rx, ry;
    load_layout<PMAX>(p, cand_xy, nobj_total, j, ob, rx, ry);
    // mjx_reset :644-657: qpos from layout, qvel = ctrl = 0, forward -> pose
    float q[R::NQ], v[R::NV], ctrl[R::NU];
#pragma unroll
    for (int k = 0; k < R::NQ; ++k) q[k] = 0.f;
#pragma unroll
    for (int k = 0; k < R::NV; ++k) v[k] = 0.f;
#pragma unroll
    for (int k = 0; k < R::NU; ++k) ctrl[k] = 0.f;
    R::place(q, rx, ry);
    float pose[4] = {rx, ry, 1.0f, 0.0f};
    world_pose(p, pose);
    float* row = tile + tid * p.D;
    build_obs_row<R, PMAX>(p, row, pose, ob, ctrl, q, v, 0.f, 0.f, 0.f, 0.f);
    if (live) {
        float oq[R::NQ], ov[R::NV], opose[4], odone, osteps;
        R::load(dyn, p.Npad, i, oq, ov, opose, odone, osteps);
        R::store(dyn, p.Npad, i, q, v, pose, odone, 0.0f); // _done kept, _steps = 0 (:463)
#pragma unroll
        for (int k = 0; k < PMAX; ++k)
            if (k < p.P) obj[(size_t)k * p.Npad + i] = ob[k];
    }
    __syncthreads();
    const int nenv = min(BLOCK, p.N - env0);
    flush_tile<BLOCK>(tile, obs + (size_t)env0 * p.D, nenv * p.D);
}

// ---------------------------------------------------------------------------
// Engine.reset_done (engine.py:497-505, mjx_reset_done :702-731)
// ---------------------------------------------------------------------------
template <class R, int BLOCK, int PMAX>
__global__ __launch_bounds__(BLOCK) void reset_done_kernel(Params p, int nobj_total, uint32_t k10,
                                                           uint32_t k11, uint32_t k20, uint32_t k21,
                                                           const int* __restrict__ layout_size,
                                                           const int* __restrict__ cand_of,
                                                           const float2* __restrict__ cand_xy,
                                                           float4* __restrict__ dyn,
                                                           float4* __restrict__ obj,
                                                           const float* obs_in, float* obs_out)
{
    extern __shared__ float4 tile4[];
    float* tile = reinterpret_cast<float*>(tile4);
    const int tid = threadIdx.x;
    const int env0 = blockIdx.x * BLOCK;
    const int i = env0 + tid;
    const bool live = i < p.N;
    const int L = *layout_size;
    float oq[R::NQ], ov[R::NV], opose[4], odone, osteps;
    R::load(dyn, p.Npad, i, oq, ov, opose, odone, osteps);
    const bool dn = live && (odone > 0.0f) && (L > 0);
    const int any = __syncthreads_or(dn ? 1 : 0);
    const int nenv = min(BLOCK, p.N - env0);
    const int total = nenv * p.D;
    if (!any && obs_in == obs_out) return; // nothing to do for this tile
    // stage the old rows (self._obs) in LDS
    stage_tile<BLOCK>(tile, obs_in + (size_t)env0 * p.D, total);
    __syncthreads();
    if (dn) {
        const uint32_t gi = (uint32_t)(p.env_offset + i);
        const uint32_t idx = randint_at(k10, k11, k20, k21, (uint32_t)p.env_total, (uint32_t)L, gi);
        const int j = cand_of[idx];
        float4 ob[PMAX];
        float rx, ry;
        load_layout<PMAX>(p, cand_xy, nobj_total, j, ob, rx, ry);
        // "fake step" (:719-724) from rest with zero ctrl leaves qpos/qvel unchanged;
        // its forward() gives pose(qpos_reset) for the obs; data keeps the STALE xpos/xmat (:731)
        float q[R::NQ], v[R::NV], ctrl[R::NU];
#pragma unroll
        for (int k = 0; k < R::NQ; ++k) q[k] = 0.f;
#pragma unroll
        for (int k = 0; k < R::NV; ++k) v[k] = 0.f;
#pragma unroll
        for (int k = 0; k < R::NU; ++k) ctrl[k] = 0.f;
        R::place(q, rx, ry);
        float pose[4] = {rx, ry, 1.0f, 0.0f};
        world_pose(p, pose);
        if (R::kRestFixed) {
            build_obs_row<R, PMAX>(p, tile + tid * p.D, pose, ob, ctrl, q, v, 0.f, 0.f, 0.f, 0.f);
        } else { // the fake step moves the robot: its qpos/qvel feed the obs only
            float fq[R::NQ], fv[R::NV], fa[R::NV];
#pragma unroll
            for (int k = 0; k < R::NQ; ++k) fq[k] = q[k];
#pragma unroll
            for (int k = 0; k < R::NV; ++k) fv[k] = 0.f;
            for (int k = 0; k < p.physics_steps; ++k) R::template substep<false>(fq, fv, ctrl, pose, fa);
            world_pose(p, pose);
            build_obs_row<R, PMAX>(p, tile + tid * p.D, pose, ob, ctrl, fq, fv, 0.f, 0.f, 0.f, 0.f);
        }
        R::store(dyn, p.Npad, i, q, v, opose, odone, osteps);
#pragma unroll
        for (int k = 0; k < PMAX; ++k)
            if (k < p.P) obj[(size_t)k * p.Npad + i] = ob[k];
    }
    __syncthreads();
    flush_tile<BLOCK>(tile, obs_out + (size_t)env0 * p.D, total);
}

// ---------------------------------------------------------------------------
// T x (Engine.step -> Engine.reset_done) for LARGE batches, thread-per-env: the state, the layout and
// the history stay in registers across the T steps of one launch, so per env-step only the action is
// read and the obs row, reward, cost and done are written (192 B for the Point task instead of 372 B);
// the layout row of a re-initialised env is fetched when its episode ends.  Same arithmetic as
// step_kernel + reset_done_kernel, hence the same bits.
// ---------------------------------------------------------------------------
template <class R, int BLOCK, int PMAX, bool kDef>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(R::kRestFixed ? (PMAX <= 9 ? 4 : 2) : 1)))
void thread_rollout_kernel(Params p_in, RolloutArgs r,
                                                               float4* __restrict__ dyn,
                                                               float4* __restrict__ obj,
                                                               float4* __restrict__ hist)
{
    const Params p = fold_params<R, kDef>(p_in);
    extern __shared__ float4 tile4[];
    float* tile = reinterpret_cast<float*>(tile4);
    const int tid = threadIdx.x;
    const int env0 = blockIdx.x * BLOCK;
    const int i = env0 + tid;
    const bool live = i < p.N;
    const int nenv = min(BLOCK, p.N - env0);

    float q[R::NQ], v[R::NV], pose0[4], done0, steps;
    R::load(dyn, p.Npad, i, q, v, pose0, done0, steps);
    float4 ob[PMAX];
#pragma unroll
    for (int k = 0; k < PMAX; ++k)
        ob[k] = (k < p.P) ? obj[(size_t)k * p.Npad + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float P1x = 0.f, P1y = 0.f, done1 = 0.f;   // last_last_data.xpos, last_last_done
    if (p.hist_on) { const float4 h = hist[i]; P1x = h.x; P1y = h.y; done1 = h.z; }
    bool touched_layout = false;
    const int L = r.do_reset ? *r.layout_size : 0; // constant for the whole launch
    // rows of the LDS tile are obs_stride wide: D, or the packed hand-off row (obs | action | reward cost done)
    const int RS = r.obs_stride;
    const bool packed = r.act_out != nullptr;
    float* row = tile + tid * RS;

    for (int t = 0; t < r.T; ++t) {
        float a[R::NA];
        load_action<R>(r.act, (size_t)t * p.N + (live ? i : 0), a);
        const bool have_last = (r.hist0 + t) >= 1, have_last_last = (r.hist0 + t) >= 2;
        const float last_done = done0, done2 = done1;
        const float L1x = pose0[0], L1y = pose0[1], P2x = P1x, P2y = P1y;

        float ctrl[R::NU];
        R::convert_action(pose0, a, ctrl); // :672-685, PRE-step xmat
        float pose[4], qacc[R::NV];
#pragma unroll
        for (int k = 0; k < R::NV; ++k) qacc[k] = 0.f;
        for (int k = 0; k < p.physics_steps; ++k) R::template substep<false>(q, v, ctrl, pose, qacc);
        world_pose(p, pose);

        float vel0 = 0.f, vel1 = 0.f, acc0 = 0.f, acc1 = 0.f;
        if (p.hist_on)
            ego_vel_acc(p, pose, L1x, L1y, P2x, P2y, last_done, done2, have_last, have_last_last, vel0, vel1, acc0,
                        acc1);
        const bool bad = build_obs_row<R, PMAX>(p, row, pose, ob, ctrl, q, v, vel0, vel1, acc0, acc1);

        // reward_done :787-802
        const float dg = dist2(ob[0].x, ob[0].y, pose[0], pose[1]);
        float last = dg;
        if (have_last && !(last_done > 0.0f)) last = dist2(ob[0].x, ob[0].y, L1x, L1y);
        const float dd = last - dg;
        float rw = dd * p.reward_distance;
        float dn = dg < p.goal_size ? 1.0f : 0.0f;
        if (fabsf(dd) > 1.0f) { dn = 1.0f; rw = 0.0f; }
        // cost :804-811
        float cs = 0.0f;
#pragma unroll
        for (int k = 0; k < PMAX; ++k) {
            if (k > 0 && 2 * k < p.nobj) cs = cs + cost_term(p, 2 * k, ob[k].x, ob[k].y, pose);
            if (2 * k + 1 < p.nobj) cs = cs + cost_term(p, 2 * k + 1, ob[k].z, ob[k].w, pose);
        }
        if (bad) { rw = 0.0f; dn = 1.0f; }           // :696-699
        if (steps > p.num_steps_f) dn = 1.0f;         // :492
        steps = dn > 0.0f ? 0.0f : steps + 1.0f;      // :493

        if (packed) {
#pragma unroll
            for (int k = 0; k < R::NA; ++k) row[p.D + k] = a[k];
            row[p.D + R::NA] = rw; row[p.D + R::NA + 1] = cs; row[p.D + R::NA + 2] = dn;
        } else if (live) {
            const size_t te = (size_t)t * p.N + i;
            r.rew[te] = rw; r.cost[te] = cs; r.done[te] = dn;
        }

        // commit the history
        P1x = L1x; P1y = L1y; done1 = last_done;
#pragma unroll
        for (int k = 0; k < 4; ++k) pose0[k] = pose[k];
        done0 = dn;

        // reset_done :497-505 for the envs that just finished
        if (r.do_reset) {
            if (live && dn > 0.0f && L > 0) {
                const uint4 kk = r.keys ? r.keys[t] : r.key0;
                const uint32_t idx = randint_at(kk.x, kk.y, kk.z, kk.w, (uint32_t)p.env_total, (uint32_t)L,
                                                (uint32_t)(p.env_offset + i));
                float rx, ry;
                load_layout<PMAX>(p, r.cand_xy, r.nobj_total, r.cand_of[idx], ob, rx, ry);
                float zc[R::NU];
#pragma unroll
                for (int k = 0; k < R::NQ; ++k) q[k] = 0.f;
#pragma unroll
                for (int k = 0; k < R::NV; ++k) v[k] = 0.f;
#pragma unroll
                for (int k = 0; k < R::NU; ++k) zc[k] = 0.f;
                R::place(q, rx, ry);
                float rpose[4] = {rx, ry, 1.0f, 0.0f};
                world_pose(p, rpose);
                if (R::kRestFixed) {
                    build_obs_row<R, PMAX>(p, row, rpose, ob, zc, q, v, 0.f, 0.f, 0.f, 0.f);
                } else { // the fake step (:719-724) moves the robot: qpos | qvel | pose tabulated with the pool
                    float fq[R::NQ], fv[R::NV];
                    const float* frow = r.fake + (size_t)idx * (R::NQ + R::NV + 4);
#pragma unroll
                    for (int k = 0; k < R::NQ; ++k) fq[k] = frow[k];
#pragma unroll
                    for (int k = 0; k < R::NV; ++k) fv[k] = frow[R::NQ + k];
#pragma unroll
                    for (int k = 0; k < 4; ++k) rpose[k] = frow[R::NQ + R::NV + k];
                    build_obs_row<R, PMAX>(p, row, rpose, ob, zc, fq, fv, 0.f, 0.f, 0.f, 0.f);
                }
                touched_layout = true;
            }
        }
        // (LDS-only synchronisation: __syncthreads() would make every wave wait, twice per step, until the tile's global
        // stores of the step before have been acknowledged -- s_waitcnt vmcnt(0) of its all-address-space fence)
        lds_sync<BLOCK>();
        flush_tile<BLOCK>(tile, r.obs + ((size_t)t * p.N + env0) * RS, nenv * RS);
        lds_sync<BLOCK>(); // the tile is rewritten by the next step
    }

    if (live) {
        R::store(dyn, p.Npad, i, q, v, pose0, done0, steps);
        if (p.hist_on) hist[i] = make_float4(P1x, P1y, done1, 0.f);
        if (touched_layout) {
#pragma unroll
            for (int k = 0; k < PMAX; ++k)
                if (k < p.P) obj[(size_t)k * p.Npad + i] = ob[k];
        }
    }
}

// ---------------------------------------------------------------------------
// Install a speculated reset_done (rd_j, written by the lane-group step with do_reset == 2) for the consumers
// that do not apply it on load: qpos/qvel/layout of the finished envs, stale pose / done / steps kept (:715-731).
// ---------------------------------------------------------------------------
template <class R, int BLOCK, int PMAX>
__global__ __launch_bounds__(BLOCK) void commit_pending_kernel(Params p, int nobj_total, int n_rows, const int* __restrict__ rd_j,
                                                               const float2* __restrict__ cand_xy,
                                                               float4* __restrict__ dyn, float4* __restrict__ obj)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= p.N) return;
    const int j = rd_j[i];
    if (j < 0 || j >= n_rows) return;
    float4 ob[PMAX];
    float rx, ry;
    load_layout<PMAX>(p, cand_xy, nobj_total, j, ob, rx, ry);
    float q[R::NQ], v[R::NV], pose0[4], done0, steps;
    R::load(dyn, p.Npad, i, q, v, pose0, done0, steps);
#pragma unroll
    for (int k = 0; k < R::NQ; ++k) q[k] = 0.f;
#pragma unroll
    for (int k = 0; k < R::NV; ++k) v[k] = 0.f;
    R::place(q, rx, ry);
    R::store(dyn, p.Npad, i, q, v, pose0, done0, steps);
#pragma unroll
    for (int k = 0; k < PMAX; ++k)
        if (k < p.P) obj[(size_t)k * p.Npad + i] = ob[k];
}

// ---------------------------------------------------------------------------
// reset_done's fake step (engine.py:719-724) for the robots that move at rest (Ant, Walker), tabulated per valid
// layout of a freshly sampled pool: one thread per row c of the compacted list runs mjx.step from the rest state at
// that layout's robot position with zero ctrl and stores qpos | qvel | pose.  The lane-group kernels read the row
// instead of stepping a second time, which leaves them ONE call site of the step -- inlined, no call ABI, no
// callee-saved registers to spill (Ant 12.7 -> 10.0 us per step, Walker 21.9 -> 18.0).  Same function as the
// thread-per-env kernels' own fake step, which equals the lane-group form bit for bit (parity tests of both families).
// ---------------------------------------------------------------------------
template <class R, int BLOCK>
__global__ __launch_bounds__(BLOCK) void fake_table_kernel(Params p, int nobj_total, const int* __restrict__ layout_size,
                                                           const int* __restrict__ cand_of,
                                                           const float2* __restrict__ cand_xy, float* __restrict__ fake)
{
    const int c = blockIdx.x * BLOCK + threadIdx.x;
    if (c >= *layout_size) return;
    const float2 rb = cand_xy[(size_t)cand_of[c] * nobj_total + nobj_total - 1];
    float q[R::NQ], v[R::NV], ctrl[R::NU], pose[4] = {rb.x, rb.y, 1.0f, 0.0f}, qacc[R::NV];
#pragma unroll
    for (int k = 0; k < R::NQ; ++k) q[k] = 0.f;
#pragma unroll
    for (int k = 0; k < R::NV; ++k) v[k] = 0.f;
#pragma unroll
    for (int k = 0; k < R::NU; ++k) ctrl[k] = 0.f;
    R::place(q, rb.x, rb.y);
    for (int k = 0; k < p.physics_steps; ++k) R::template substep<false>(q, v, ctrl, pose, qacc);
    world_pose(p, pose);
    float* row = fake + (size_t)c * (R::NQ + R::NV + 4);
#pragma unroll
    for (int k = 0; k < R::NQ; ++k) row[k] = q[k];
#pragma unroll
    for (int k = 0; k < R::NV; ++k) row[R::NQ + k] = v[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) row[R::NQ + R::NV + k] = pose[k];
}

// ---------------------------------------------------------------------------
// Lane-group kernel for SMALL batches (latency regime, env_num ~ 10^3..10^4).
//
// 16 lanes cooperate on one environment, 4 environments per wave64, one wave per
// workgroup: env_num=2000 becomes 500 single-wave workgroups spread over the
// chip instead of 32 waves each grinding through 9 objects serially.
//   * every lane carries the env's dynamic state and integrates it (redundantly:
//     identical operations, identical bits) -- no broadcast on the critical path;
//   * lane o evaluates object o (goal, hazard0..): ego vector, sqrt, atan2, exp,
//     alias, hazard cost term -- the 9 transcendental chains run side by side;
//   * the per-object (bin, sensor, a1, a2) records are exchanged through 1 KiB of
//     LDS and lane b folds them into lidar bin b (the scatter-max becomes a
//     gather-max with the same operand order, so results are bit-identical to the
//     thread-per-env kernel);
//   * the kernel is persistent over T steps: state and layout stay in registers,
//     per step it reads 8 B of action and writes the obs row + 3 scalars;
//   * reset_done (engine.py:497-505) is folded in: a wave-uniform ballot of the done
//     flags gates the re-draw of the layout index and the rebuild of the obs row.
// ---------------------------------------------------------------------------

constexpr int kGL = 16; // lanes per environment

// Stamps INSIDE the step loop of the lane-group kernel (3: step begins, 4: dynamics done, 5: observation done, 6: step
// ends) are compiled in only with -DGX_LOOP_STAMPS: the store of a stamp sits behind a branch, and at the merge point the
// compiler waits for it whether it happened or not -- an s_waitcnt vmcnt(0) per stamp and step, i.e. every step of the
// closed-loop kernels waited for its own output stores (found in round 5 in the ISA of group_rollout_kernel<.., 2>).
#ifdef GX_LOOP_STAMPS
constexpr bool kLoopStamps = true;
#else
constexpr bool kLoopStamps = false;
#endif
// profiling aid: shader-clock stamp k of this workgroup (RolloutArgs::stamps, normally null)
GX_D void stamp(const RolloutArgs& r, int k)
{
    if (r.stamps && threadIdx.x == 0) r.stamps[(size_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime();
}

// Workgroup synchronisation of the lane-group kernels.  With ONE wave per workgroup (BT == 64) the LDS
// operations of the wave execute in program order, so the exchange through LDS only needs the compiler
// to keep that order: wave-scope fences cost no instruction.  __syncthreads() would also wait for every
// outstanding global store (s_waitcnt vmcnt(0) of the workgroup-scope release fence), i.e. for the
// observation rows of the previous step to be acknowledged by L2 -- on the critical path of every step.
// kWave: only for the robots whose step is inline register code (Point, Swimmer).  The Ant / Walker kernels call
// their step as a real function with stack arrays in scratch and keep the full barrier.
template <int BT, bool kWave>
GX_D void group_sync()
{
    // (any workgroup size: what is exchanged here belongs to ONE env = 16 lanes of one wave; the policy kernels' four-wave
    // workgroups took __syncthreads() -- an s_waitcnt vmcnt(0) on the step's output stores -- five times per step until round 5)
    if (kWave) { // LDS-only fences: a fence over all address spaces, even at wave scope, makes the compiler wait for vmcnt(0)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    } else {
        __syncthreads();
    }
}
template <int BT, bool kWave>
GX_D bool group_any(bool v)
{
    if (kWave) return __ballot(v) != 0ull;
    return __syncthreads_or(v ? 1 : 0) != 0;
}

template <int OPL, int BPL>
struct GroupObs { float gl[BPL], hl[BPL], pl[BPL], comp0, comp1, cost; bool bad; };

// LDS of the lane-group kernels: per-object records (exact path), hazard cost terms, and the lidar bins of
// the (goal, hazards, pillars) rows of every env of the workgroup (scatter-max path)
template <int OPL, int BPL, int BT>
struct GroupLds {
    float4 rec[OPL][BT];
    float term[OPL][BT];
    int bins[3][BT / kGL][kGL * BPL];
};

// object phase + LDS exchange + bin phase for one pose
// `lane` = thread index in the workgroup (BT threads = BT/16 environments); must be reached by the
// whole workgroup.
//
// Lidar bins: the reference scatter-maxes every object's (sensor, alias) values into its row one object after
// the other (engine.py:872-899).  All contributions are >= +0 or negative numbers that lose against the
// initial 0, and the maximum of such values does not depend on the order, so the normal case is an LDS
// scatter-max on the BIT PATTERNS (ds_max_i32: signed integer order == float order for non-negative floats,
// every negative float is a negative integer): 3 atomics per object instead of a 3-compare / 3-select chain per
// (object, bin) pair.  If any value that would land in an observation is NaN or Inf (wave-uniform test; the
// NaN guard of engine.py:696-699 then ends the episode) the workgroup takes the exact path, which replays the
// reference's NaN-propagating sequential maxima through per-object records.
template <int OPL, int BPL, int BT, bool kWave>
GX_D GroupObs<OPL, BPL> group_observe(const Params& p, GroupLds<OPL, BPL, BT>& S, int lane,
                                      const float (&pose)[4], float gx, float gy,
                                      const float (&ox)[OPL], const float (&oy)[OPL])
{
    const int l = lane & (kGL - 1), gbase = lane & ~(kGL - 1), g = lane >> 4;
    const int B = p.bins;
    GroupObs<OPL, BPL> out;
    bool bad = false;
    LidarTerms tt[OPL];
    int rowof[OPL];
#pragma unroll
    for (int j = 0; j < OPL; ++j) {
        const int o = l + kGL * j;
        const bool valid = o < p.nobj;
        tt[j] = lidar_terms(p, ox[j], oy[j], pose);
        const int cls = (o == 0) ? 0 : (o <= p.H ? 1 : 2);
        const bool enabled = (cls == 0) ? (p.off_gl >= 0) : (cls == 1 ? p.off_hl >= 0 : p.off_pl >= 0);
        rowof[j] = (valid && enabled) ? cls : -1;
        if (valid && enabled) bad = bad || lidar_bad(p, tt[j]);
        S.term[j][lane] = (valid && o >= 1) ? cost_term(p, o, ox[j], oy[j], pose) : 0.0f;
    }
#pragma unroll
    for (int jb = 0; jb < BPL; ++jb) {
        const int b = l + kGL * jb;
        S.bins[0][g][b] = 0; S.bins[1][g][b] = 0; S.bins[2][g][b] = 0;
    }
    const bool exact = group_any<BT, kWave>(bad); // includes the synchronisation of the zeroed bins
    if (!exact) {
        group_sync<BT, kWave>();
#pragma unroll
        for (int j = 0; j < OPL; ++j) {
            if (rowof[j] >= 0) {
                int* rowp = S.bins[rowof[j]][g];
                const LidarTerms& t = tt[j];
                if (t.bin < B) atomicMax(&rowp[t.bin], __float_as_int(t.sensor));
                if (p.lidar_alias) {
                    atomicMax(&rowp[bin_plus(t.bin, B)], __float_as_int(t.a1));
                    atomicMax(&rowp[bin_minus(t.bin, B)], __float_as_int(t.a2));
                }
            }
        }
        group_sync<BT, kWave>();
#pragma unroll
        for (int jb = 0; jb < BPL; ++jb) {
            const int b = l + kGL * jb;
            out.gl[jb] = __int_as_float(S.bins[0][g][b]);
            out.hl[jb] = __int_as_float(S.bins[1][g][b]);
            out.pl[jb] = __int_as_float(S.bins[2][g][b]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < OPL; ++j) {
            LidarTerms t = tt[j];
            if (rowof[j] < 0) { t.bin = -1000; t.sensor = 0.f; t.a1 = 0.f; t.a2 = 0.f; }
            S.rec[j][lane] = make_float4(__int_as_float(t.bin), t.sensor, t.a1, t.a2);
        }
        group_sync<BT, kWave>();
#pragma unroll
        for (int jb = 0; jb < BPL; ++jb) {
            const int b = l + kGL * jb;
            float gl = 0.0f, hl = 0.0f, pl = 0.0f;
            for (int o = 0; o < p.nobj; ++o) {
                const float4 rc = S.rec[o >> 4][gbase + (o & 15)];
                LidarTerms t;
                t.bin = __float_as_int(rc.x); t.sensor = rc.y; t.a1 = rc.z; t.a2 = rc.w;
                const float c = lidar_contrib(p, t, b);
                if (o == 0) gl = nmax(gl, c);
                else if (o <= p.H) hl = nmax(hl, c);
                else pl = nmax(pl, c);
            }
            out.gl[jb] = gl; out.hl[jb] = hl; out.pl[jb] = pl;
        }
    }
    float cs = 0.0f;
#pragma unroll 8
    for (int o = 1; o < p.nobj; ++o) cs = cs + S.term[o >> 4][gbase + (o & 15)];
    out.cost = cs;
    { // obs_compass :834-844 (same expression as ego_xy of the goal)
        const float dx = gx - pose[0], dy = gy - pose[1];
        out.comp0 = dx * pose[2] + dy * pose[3];
        out.comp1 = dx * (-pose[3]) + dy * pose[2];
        if (p.off_comp >= 0) bad = bad || notfinite(out.comp0) || notfinite(out.comp1);
    }
    // any lane of this env's group
    const unsigned long long m = __ballot(bad);
    out.bad = ((m >> (gbase & 63)) & 0xFFFFull) != 0ull;
    group_sync<BT, kWave>();
    return out;
}

// value k of a small register array selected by a per-lane index (compare-select chain).  Every element goes
// through an empty asm first: left alone, the optimiser folds the chain back into a dynamically indexed load,
// the array then cannot stay in registers, AMDGPUPromoteAlloca moves it to LDS, and its per-lane LDS slot is
// addressed with the workgroup sizes read from the AQL dispatch packet -- a scalar load from HOST memory
// (~12 us per launch, measured with the in-kernel stamps: 24 k cycles between "state arrived" and the first
// step of every launch of the round-1 kernels).
template <int N>
GX_D float pick(const float (&a)[N], int k)
{
    float r = a[0];
    asm volatile("" : "+v"(r));
#pragma unroll
    for (int i = 1; i < N; ++i) {
        float ai = a[i];
        asm volatile("" : "+v"(ai));
        r = (k == i) ? ai : r;
    }
    return r;
}

// kPol: 0 open loop (action tape), 1 policy evaluated with VALU fmaf chains (one wave per workgroup),
//       2 policy evaluated with fp32 MFMA tiles (four waves = 16 envs per workgroup), width 64, weights in LDS
//       3 the same for width 128: hidden-layer weights resident in REGISTERS as the lanes' MFMA B operands (gx_policy.h)
//       192 / 256 that width, hidden-layer weights STREAMED from the L2-resident transposed copy (gx_policy.h)
template <class R, int OPL, int BPL, bool kQacc, bool kDef, int kPol>
__global__ __launch_bounds__(kPol >= 2 ? 256 : 64) void group_rollout_kernel(Params p_in, RolloutArgs r,
                                                                            PolicyArgs pol,
                                                                            float4* __restrict__ dyn,
                                                                            float4* __restrict__ obj,
                                                                            float4* __restrict__ hist)
{
    constexpr int BT = (kPol >= 2) ? 256 : 64;
    constexpr bool kPolicy = kPol != 0;
    // k-steps of the first hidden layer in the register-resident form: the robot's default-task observation width
    // (qpos, qvel, ctrl, compass, two 16-bin lidars), padded to fours -- the launcher checks that p.D matches
    constexpr int KS1 = (R::NQ + R::NV + R::NU + 2 + 32 + 3) / 4;
    constexpr bool kStream = kPol == 192 || kPol == 256;   // streaming form of that width
    constexpr int HW = kStream ? kPol : kPolHd2, HWS = HW + 4;
    const Params p = fold_params<R, kDef>(p_in);
    stamp(r, 0);
    __shared__ GroupLds<OPL, BPL, BT> S;
    extern __shared__ float4 pol_lds4[];
    const int lane = threadIdx.x;           // thread in the workgroup
    const int l = lane & (kGL - 1);         // lane within the env's 16-lane group
    const int env = blockIdx.x * (BT / kGL) + (lane >> 4);
    const bool live = env < p.N;
    const int e = live ? env : 0;

    // ---- policy: weights into LDS, entry observation into this env's LDS row
    float* pol_lds = reinterpret_cast<float*>(pol_lds4);
    MlpLds wpi, wv;
    Mlp2Head hpi, hvv;
    Pol2Regs<kPol == 3 ? KS1 : 1> PW;
    float *xrow = nullptr, *hbuf = nullptr, *X = nullptr, *H1 = nullptr, *H2 = nullptr;
    int XS = 0;
    float pstd[R::NA], plstd[R::NA];
    if constexpr (kPol == 3 || kStream) {
        const int D = p.D, A = R::NA;
        float* pi_img = pol_lds;
        float* v_img = pi_img + pad4(mlp2_head_floats(A, HW));
        float* ls_img = v_img + pad4(mlp2_head_floats(1, HW));
        XS = pad4(D) + 1;
        X = ls_img + pad4(2 * A);
        H1 = X + pad4(16 * XS);
        H2 = H1 + 2 * 16 * HWS;
        xrow = X + (lane >> 4) * XS;
        for (int i = lane; i < 16 * XS; i += BT) X[i] = 0.0f; // zero padding columns
        if constexpr (kPol == 3) pol2_load<KS1>(PW, pol.params, D, A, lane >> 6, lane & 63);
        mlp2_head_stage(pi_img, pol.params, D, A, lane, BT, HW);
        mlp2_head_stage(v_img, pol.params + mlp2_floats(D, A, HW), D, 1, lane, BT, HW);
        hpi = mlp2_head_view(pi_img, A, HW);
        hvv = mlp2_head_view(v_img, 1, HW);
        const float* gls = pol.params + mlp2_floats(D, A, HW) + mlp2_floats(D, 1, HW);
#pragma unroll
        for (int d = 0; d < A; ++d) {
            pstd[d] = exp_f(gls[d]);      // std = exp(log_std)          trpo_core.py:123
            plstd[d] = log_f(pstd[d]);    // torch.log(pi.stddev)        trpo_core.py:173
            if (blockIdx.x == 0 && lane == d) pol.logstd[d] = plstd[d];
        }
        __syncthreads();
        for (int k = l; k < D; k += kGL) xrow[k] = pol.obs0[(size_t)e * D + k];
        __syncthreads();
    } else if (kPolicy) {
        const int D = p.D, A = R::NA;
        const int rows = (kPol == 2) ? pad4(D) : D;
        float* pi_img = pol_lds;
        float* v_img = pi_img + pad4(mlp_lds_floats(rows, A));
        float* ls_img = v_img + pad4(mlp_lds_floats(rows, 1));
        float* rest = ls_img + pad4(2 * A);
        if (kPol == 2) {
            XS = pad4(D) + 1;
            X = rest;
            H1 = X + pad4(16 * XS);
            H2 = H1 + 2 * 16 * kPolHS;
            xrow = X + (lane >> 4) * XS;
            hbuf = H1 + (lane >> 4) * kPolHS; // scratch for the final critic pass
            for (int i = lane; i < 16 * XS; i += BT) X[i] = 0.0f; // zero padding columns
        } else {
            hbuf = rest + (lane >> 4) * 2 * kPolHd;
            xrow = rest + 4 * 2 * kPolHd + (lane >> 4) * pad4(D);
        }
        mlp_stage(pi_img, pol.params, D, rows, A, lane, BT);
        mlp_stage(v_img, pol.params + mlp_floats(D, A), D, rows, 1, lane, BT);
        wpi = mlp_lds_view(pi_img, rows, A);
        wv = mlp_lds_view(v_img, rows, 1);
        const float* gls = pol.params + mlp_floats(D, A) + mlp_floats(D, 1);
#pragma unroll
        for (int d = 0; d < A; ++d) {
            pstd[d] = exp_f(gls[d]);      // std = exp(log_std)          trpo_core.py:123
            plstd[d] = log_f(pstd[d]);    // torch.log(pi.stddev)        trpo_core.py:173
            if (blockIdx.x == 0 && lane == d) pol.logstd[d] = plstd[d];
        }
        __syncthreads();
        for (int k = l; k < D; k += kGL) xrow[k] = pol.obs0[(size_t)e * D + k];
        __syncthreads();
    }

    // ---- state (every lane of the group holds a copy)
    float q[R::NQ], v[R::NV], pose0[4], done0, steps;
    R::load(dyn, p.Npad, e, q, v, pose0, done0, steps);
    float P1x = 0.f, P1y = 0.f, done1 = 0.f;
    if (p.hist_on) { const float4 h = hist[e]; P1x = h.x; P1y = h.y; done1 = h.z; }
    const float2* obj2 = reinterpret_cast<const float2*>(obj);
    float ox[OPL], oy[OPL];
#pragma unroll
    for (int j = 0; j < OPL; ++j) {
        const int o = l + kGL * j;
        float2 t = make_float2(0.f, 0.f);
        if (o < p.nobj) t = obj2[((size_t)(o >> 1) * p.Npad + e) * 2 + (o & 1)];
        ox[j] = t.x; oy[j] = t.y;
    }
    float gx, gy;
    { const float2 g = obj2[(size_t)e * 2]; gx = g.x; gy = g.y; }
    bool touched_layout = false;
    if (r.commit) { // the reset_done the previous launch speculated was requested: install it (engine.py:715-718)
        const int jj = r.rd_j[e];
        if (live && jj >= 0 && jj < r.n_rows) {
            const float2* rowp = r.cand_xy + (size_t)jj * r.nobj_total;
#pragma unroll
            for (int j = 0; j < OPL; ++j) {
                const int o = l + kGL * j;
                if (o < p.nobj) { const float2 t2 = rowp[o]; ox[j] = t2.x; oy[j] = t2.y; }
            }
            const float2 g = rowp[0], rb = rowp[r.nobj_total - 1];
            gx = g.x; gy = g.y;
#pragma unroll
            for (int k = 0; k < R::NQ; ++k) q[k] = 0.f;
#pragma unroll
            for (int k = 0; k < R::NV; ++k) v[k] = 0.f;
            R::place(q, rb.x, rb.y);
            touched_layout = true;
        }
    }
    const int L = r.do_reset ? *r.layout_size : 0; // constant for the whole launch

    float a_next[R::NA];
#pragma unroll
    for (int d = 0; d < R::NA; ++d) a_next[d] = 0.f;
    if (!kPolicy) load_action<R>(r.act, (size_t)e, a_next);
    if (r.stamps) { // profiling: when have the state and the first action arrived?
        stamp(r, 1);
        float s_ = q[0] + v[0] + pose0[0] + ox[0] + a_next[0] + gx;
        asm volatile("" ::"v"(s_));
        stamp(r, 2);
    }
    if (kPolicy) {
        // Everything loaded before the step loop is consumed HERE once.  Left to the compiler, the wait for a value first
        // used inside the loop (the goal, the objects) is placed at that use -- inside the loop, as s_waitcnt vmcnt(0), where
        // from the second step on it only waits for the previous step's output STORES (loads and stores share the counter).
        float s_ = (float)L + gx + gy + done0 + steps + P1x + P1y + done1;
#pragma unroll
        for (int k = 0; k < R::NQ; ++k) s_ += q[k];
#pragma unroll
        for (int k = 0; k < R::NV; ++k) s_ += v[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) s_ += pose0[k];
#pragma unroll
        for (int j = 0; j < OPL; ++j) s_ += ox[j] + oy[j];
        asm volatile("" ::"v"(s_));
    }
    const int tstar = r.T > 100 ? 100 : r.T - 1; // the step whose phases are stamped (profiling aid)
    for (int t = 0; t < r.T; ++t) {
        if (kLoopStamps && t == tstar) stamp(r, 3);
        float a[R::NA];
#pragma unroll
        for (int d = 0; d < R::NA; ++d) a[d] = a_next[d];
        if (!kPolicy) {
            if (t + 1 < r.T) load_action<R>(r.act, (size_t)(t + 1) * p.N + e, a_next);
        } else {
            // ac.step(o): a ~ N(mu(o), std), logp, v(o)   trpo_core.py:166-173
            const size_t te = (size_t)t * p.N + env;
            float mu[R::NA], vv[1];
            if constexpr (kPol == 3 || kStream) {
                if constexpr (kPol == 3) pol2_hidden<KS1>(PW, X, XS, H1, H2, lane >> 6, lane & 63);
                else polS_hidden<HW>(pol.wt, hpi, hvv, X, XS, pad4(p.D), H1, H2, lane >> 6, lane & 63);
                const float* h2p = H2 + (lane >> 4) * HWS;
#pragma unroll
                for (int o = 0; o < R::NA; ++o) mu[o] = head2_out<HW>(hpi, o, l, h2p);
                vv[0] = head2_out<HW>(hvv, 0, l, h2p + 16 * HWS);
            } else if (kPol == 2) {
                mfma_hidden(wpi, wv, X, XS, pad4(p.D), H1, H2, lane >> 6, lane & 63);
                const float* h2p = H2 + (lane >> 4) * kPolHS + 4 * l;
                const float4 hp = *reinterpret_cast<const float4*>(h2p);
                const float4 hc = *reinterpret_cast<const float4*>(h2p + 16 * kPolHS);
#pragma unroll
                for (int o = 0; o < R::NA; ++o) mu[o] = head_out(wpi, o, l, hp);
                vv[0] = head_out(wv, 0, l, hc);
            } else {
                actor_critic_forward<R::NA>(wpi, wv, xrow, hbuf, p.D, l, mu, vv[0]);
            }
            float z[R::NA];
#pragma unroll
            for (int j = 0; j < R::NA / 2; ++j) // one counter per pair of action dimensions
                normal_pair(pol.seed0, pol.seed1, (uint32_t)(p.env_offset + env), (pol.t0 + (uint32_t)t) * 16u + (uint32_t)j,
                            z[2 * j], z[2 * j + 1]);
            float act[R::NA], lp = 0.0f;
#pragma unroll
            for (int d = 0; d < R::NA; ++d) {
                act[d] = fmaf(pstd[d], z[d], mu[d]);
                const float df = act[d] - mu[d];
                const float var = pstd[d] * pstd[d];
                lp = lp + ((-(df * df) / (2.0f * var) - plstd[d]) - 0.9189385332046727f);
            }
#pragma unroll
            for (int d = 0; d < R::NA; ++d) a[d] = act[d];
            if (live) {
                for (int k = l; k < p.D; k += kGL) pol.obs_in[te * p.D + k] = xrow[k];
                if (l < R::NA) {
                    pol.act[te * R::NA + l] = pick(act, l);
                    pol.mu[te * R::NA + l] = pick(mu, l);
                }
                if (l == 0) { pol.logp[te] = lp; pol.val[te] = vv[0]; }
            }
            wg_sync_lds(); // xrow is rewritten at the end of this step
        }
        const bool have_last = (r.hist0 + t) >= 1, have_last_last = (r.hist0 + t) >= 2;

        // update_data :426-431 (history shift)
        const float done2 = done1;
        const float last_done = done0;
        const float P2x = P1x, P2y = P1y;
        const float L1x = pose0[0], L1y = pose0[1]; // last_data.xpos

        // convert_action :672-685, mjx.step :689
        float ctrl[R::NU];
        R::convert_action(pose0, a, ctrl);
        float pose[4], qacc[R::NV];
#pragma unroll
        for (int k = 0; k < R::NV; ++k) qacc[k] = 0.f;
        for (int k = 0; k < p.physics_steps; ++k) group_substep<R, kQacc>(q, v, ctrl, pose, qacc, l);
        world_pose(p, pose);
        if (kLoopStamps && r.stamps && t == tstar) { asm volatile("" ::"v"(pose[0] + pose[3] + q[2])); stamp(r, 4); }

        float vel0 = 0.f, vel1 = 0.f, acc0 = 0.f, acc1 = 0.f;
        if (p.hist_on)
            ego_vel_acc(p, pose, L1x, L1y, P2x, P2y, last_done, done2, have_last, have_last_last, vel0, vel1,
                        acc0, acc1);

        GroupObs<OPL, BPL> ob = group_observe<OPL, BPL, BT, R::kRestFixed>(p, S, lane, pose, gx, gy, ox, oy);
        if (kLoopStamps && r.stamps && t == tstar) { asm volatile("" ::"v"(ob.gl[0] + ob.hl[0] + ob.cost)); stamp(r, 5); }
        bool bad = ob.bad;
        if (p.off_acc >= 0) bad = bad || notfinite(acc0) || notfinite(acc1);
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
        if (p.off_vel >= 0) bad = bad || notfinite(vel0) || notfinite(vel1);

        // reward_done :787-802
        const float dg = dist2(gx, gy, pose[0], pose[1]);
        float last = dg;
        if (have_last && !(last_done > 0.0f)) last = dist2(gx, gy, L1x, L1y);
        const float dd = last - dg;
        float rw = dd * p.reward_distance;
        float dn = dg < p.goal_size ? 1.0f : 0.0f;
        if (fabsf(dd) > 1.0f) { dn = 1.0f; rw = 0.0f; }
        if (bad) { rw = 0.0f; dn = 1.0f; }          // :696-699
        if (steps > p.num_steps_f) dn = 1.0f;        // :492
        steps = dn > 0.0f ? 0.0f : steps + 1.0f;     // :493

        // commit the history
        P1x = L1x; P1y = L1y; done1 = last_done;
#pragma unroll
        for (int k = 0; k < 4; ++k) pose0[k] = pose[k];
        done0 = dn;

        // values of this env's obs row
        float o_ctrl[R::NU], o_q[R::NQ], o_v[R::NV];
#pragma unroll
        for (int k = 0; k < R::NU; ++k) o_ctrl[k] = ctrl[k];
#pragma unroll
        for (int k = 0; k < R::NQ; ++k) o_q[k] = q[k];
#pragma unroll
        for (int k = 0; k < R::NV; ++k) o_v[k] = v[k];
        float o_v0 = vel0, o_v1 = vel1, o_a0 = acc0, o_a1 = acc1;

        // one observation row (engine.py:773-777 order), 16 lanes of the env's group writing side by side
        auto write_row = [&](float* row, const GroupObs<OPL, BPL>& gob) {
#pragma unroll
            for (int jb = 0; jb < BPL; ++jb) {
                const int b = l + kGL * jb;
                if (b < p.bins) {
                    if (p.off_gl >= 0) row[p.off_gl + b] = gob.gl[jb];
                    if (p.off_hl >= 0) row[p.off_hl + b] = gob.hl[jb];
                    if (p.off_pl >= 0) row[p.off_pl + b] = gob.pl[jb];
                }
            }
            if (l < R::NU && p.off_ctrl >= 0) row[p.off_ctrl + l] = pick(o_ctrl, l);
            if (l < R::NQ && p.off_qpos >= 0) row[p.off_qpos + l] = pick(o_q, l);
            if (l < R::NV && p.off_qvel >= 0) row[p.off_qvel + l] = pick(o_v, l);
            if (l < 2) {
                if (p.off_comp >= 0) row[p.off_comp + l] = (l == 0) ? gob.comp0 : gob.comp1;
                if (p.off_vel >= 0) row[p.off_vel + l] = (l == 0) ? o_v0 : o_v1;
                if (p.off_acc >= 0) row[p.off_acc + l] = (l == 0) ? o_a0 : o_a1;
            }
        };
        const size_t te = (size_t)t * p.N + env;
        if (live) {
            if (kQacc && l < R::NV) r.qacc[te * R::NV + l] = pick(qacc, l);
            if (r.act_out && l < R::NA) r.act_out[te * r.obs_stride + l] = pick(a, l);
            if (l == 0) {
                r.rew[te * r.sc_stride] = rw; r.cost[te * r.sc_stride] = ob.cost; r.done[te * r.sc_stride] = dn;
            }
            // Engine.step's own observation (the learner's next_o); the row after reset_done goes to obs_rd
            if (r.do_reset == 2) write_row(r.obs + te * r.obs_stride, ob);
        }

        // reset_done :497-505 folded in (workgroup-uniform gate).  do_reset == 1: installed at once (the
        // learner loop calls it whenever an env is done); do_reset == 2: only its observation and the layout
        // row it draws are recorded, the state changes when reset_done() is actually called (r.commit).
        if (r.do_reset) {
            const bool rs = live && dn > 0.0f && L > 0;
            int jrow = -1;
            if (group_any<BT, R::kRestFixed>(rs)) {
                float nox[OPL], noy[OPL], ngx = gx, ngy = gy, rx = 0.f, ry = 0.f;
                uint32_t fidx = 0; // row of the compacted layout list (Pool::fake)
#pragma unroll
                for (int j = 0; j < OPL; ++j) { nox[j] = ox[j]; noy[j] = oy[j]; }
                if (rs) {
                    const uint4 kk = r.keys ? r.keys[t] : r.key0;
                    const uint32_t idx = randint_at(kk.x, kk.y, kk.z, kk.w, (uint32_t)p.env_total, (uint32_t)L,
                                                    (uint32_t)(p.env_offset + env));
                    jrow = r.cand_of[idx];
                    fidx = idx;
                    const float2* rowp = r.cand_xy + (size_t)jrow * r.nobj_total;
#pragma unroll
                    for (int j = 0; j < OPL; ++j) {
                        const int o = l + kGL * j;
                        if (o < p.nobj) { const float2 t2 = rowp[o]; nox[j] = t2.x; noy[j] = t2.y; }
                    }
                    const float2 g = rowp[0], rb = rowp[r.nobj_total - 1];
                    ngx = g.x; ngy = g.y; rx = rb.x; ry = rb.y;
                }
                float rpose[4] = {rx, ry, 1.0f, 0.0f};
                world_pose(p, rpose);
                float fq[R::NQ], fv[R::NV];
#pragma unroll
                for (int k = 0; k < R::NQ; ++k) fq[k] = 0.f;
#pragma unroll
                for (int k = 0; k < R::NV; ++k) fv[k] = 0.f;
                R::place(fq, rx, ry);
                if (!R::kRestFixed && rs) { // the fake step (:719-724) moves the robot: its qpos/qvel feed the obs only
                    const float* frow = r.fake + (size_t)fidx * (R::NQ + R::NV + 4); // tabulated with the pool
#pragma unroll
                    for (int k = 0; k < R::NQ; ++k) fq[k] = frow[k];
#pragma unroll
                    for (int k = 0; k < R::NV; ++k) fv[k] = frow[R::NQ + k];
#pragma unroll
                    for (int k = 0; k < 4; ++k) rpose[k] = frow[R::NQ + R::NV + k];
                }
                const GroupObs<OPL, BPL> rob = group_observe<OPL, BPL, BT, R::kRestFixed>(p, S, lane, rpose, ngx, ngy, nox, noy);
                if (rs) {
                    if (r.do_reset == 1) {
#pragma unroll
                        for (int j = 0; j < OPL; ++j) { ox[j] = nox[j]; oy[j] = noy[j]; }
                        gx = ngx; gy = ngy;
#pragma unroll
                        for (int k = 0; k < R::NQ; ++k) q[k] = 0.f;
#pragma unroll
                        for (int k = 0; k < R::NV; ++k) v[k] = 0.f;
                        R::place(q, rx, ry);
                        touched_layout = true;
                    }
#pragma unroll
                    for (int k = 0; k < R::NQ; ++k) o_q[k] = fq[k];
#pragma unroll
                    for (int k = 0; k < R::NV; ++k) o_v[k] = fv[k];
#pragma unroll
                    for (int k = 0; k < R::NU; ++k) o_ctrl[k] = 0.f;
#pragma unroll
                    for (int jb = 0; jb < BPL; ++jb) { ob.gl[jb] = rob.gl[jb]; ob.hl[jb] = rob.hl[jb]; ob.pl[jb] = rob.pl[jb]; }
                    ob.comp0 = rob.comp0; ob.comp1 = rob.comp1;
                    o_v0 = o_v1 = o_a0 = o_a1 = 0.f;
                }
            }
            if (r.do_reset == 2 && live && l == 0) r.rd_j[env] = jrow;
        }

        if (live || kPolicy) {
            // closed loop: the post-reset row is the policy's next input (LDS); open loop: global
            float* row = kPolicy ? xrow : (r.do_reset == 2 ? r.obs_rd + (size_t)env * p.D : r.obs + te * r.obs_stride);
            write_row(row, ob);
        }
        if (kPolicy) wg_sync_lds();
        if (kLoopStamps && t == tstar) stamp(r, 6);
    }

    if (kPolicy) { // bootstrap inputs: o_T and V(o_T)   trpo.py:523-529
        float vlast;
        if constexpr (kPol == 3 || kStream) {
            if constexpr (kPol == 3) pol2_hidden<KS1>(PW, X, XS, H1, H2, lane >> 6, lane & 63);
            else polS_hidden<HW>(pol.wt, hpi, hvv, X, XS, pad4(p.D), H1, H2, lane >> 6, lane & 63);
            vlast = head2_out<HW>(hvv, 0, l, H2 + (16 + (lane >> 4)) * HWS);
        } else {
            vlast = critic_forward(wv, xrow, hbuf, p.D, l);
        }
        if (live) {
            for (int k = l; k < p.D; k += kGL) pol.obs_last[(size_t)env * p.D + k] = xrow[k];
            if (l == 0) pol.val_last[env] = vlast;
        }
    }

    if (live) {
        if (l == 0) {
            R::store(dyn, p.Npad, env, q, v, pose0, done0, steps);
            if (p.hist_on) hist[env] = make_float4(P1x, P1y, done1, 0.f);
        }
        if (touched_layout) {
            float2* objw = reinterpret_cast<float2*>(obj);
#pragma unroll
            for (int j = 0; j < OPL; ++j) {
                const int o = l + kGL * j;
                if (o < p.nobj) objw[((size_t)(o >> 1) * p.Npad + env) * 2 + (o & 1)] = make_float2(ox[j], oy[j]);
            }
        }
    }
    stamp(r, 7);
}

} // namespace gx
#include "gx_split_rollout.inl"
namespace gx {

// ---------------------------------------------------------------------------
// launchers (per robot)
// ---------------------------------------------------------------------------
template <class R, int BLOCK, int PMAX>
static void launch_step_bp(const Params& p, const DevBuffers& b, const float* act, float* obs,
                           float* rew, float* cost, float* done, float* qacc, hipStream_t s)
{
    const dim3 grid((p.N + BLOCK - 1) / BLOCK), blk(BLOCK);
    const size_t lds = step_lds_bytes(p, BLOCK);
    const float* a2 = act;
    if (PMAX == 5 && is_default_layout<R>(p)) {
        if (qacc)
            hipLaunchKernelGGL((step_kernel<R, BLOCK, 5, true, true>), grid, blk, lds, s, p, a2, b.dyn, b.obj, b.hist,
                               obs, rew, cost, done, qacc);
        else
            hipLaunchKernelGGL((step_kernel<R, BLOCK, 5, false, true>), grid, blk, lds, s, p, a2, b.dyn, b.obj, b.hist,
                               obs, rew, cost, done, qacc);
    } else if (qacc)
        hipLaunchKernelGGL((step_kernel<R, BLOCK, PMAX, true, false>), grid, blk, lds, s, p, a2, b.dyn, b.obj, b.hist,
                           obs, rew, cost, done, qacc);
    else
        hipLaunchKernelGGL((step_kernel<R, BLOCK, PMAX, false, false>), grid, blk, lds, s, p, a2, b.dyn, b.obj, b.hist,
                           obs, rew, cost, done, qacc);
}

// workgroup size x object-pair capacity.  256-thread workgroups are a tuning option (GX_BLOCK=256) kept for
// the light robots only: the Ant step is ~8k instructions and every extra instantiation costs build time.
#define GX_DISPATCH_P(R, FN, BLK, ...)                                 \
    do {                                                               \
        if (p.P <= 5) FN<R, BLK, 5>(__VA_ARGS__);                      \
        else if (p.P <= 9) FN<R, BLK, 9>(__VA_ARGS__);                 \
        else FN<R, BLK, 33>(__VA_ARGS__);                              \
    } while (0)
#define GX_DISPATCH_BP_R(R, FN, ...)                                   \
    do {                                                               \
        if constexpr (R::kRestFixed) {                                 \
            if (pick_block(p) == 256) { GX_DISPATCH_P(R, FN, 256, __VA_ARGS__); break; } \
        }                                                              \
        GX_DISPATCH_P(R, FN, 64, __VA_ARGS__);                         \
    } while (0)

template <class R, int BLOCK, int PMAX>
static void launch_reset_apply_bp(const Params& p, const DevBuffers& b, int nobj_total, uint32_t k10,
                                  uint32_t k11, uint32_t k20, uint32_t k21, float* obs, int* host_ls,
                                  hipStream_t s)
{
    const dim3 grid((p.N + BLOCK - 1) / BLOCK), blk(BLOCK);
    hipLaunchKernelGGL((reset_apply_kernel<R, BLOCK, PMAX>), grid, blk, step_lds_bytes(p, BLOCK), s, p,
                       nobj_total, k10, k11, k20, k21, b.pool.layout_size, b.pool.cand_of, b.pool.cand_xy, b.dyn, b.obj,
                       obs, host_ls);
}


template <class R, int BLOCK, int PMAX>
static void launch_reset_done_bp(const Params& p, const DevBuffers& b, int nobj_total, uint32_t k10,
                                 uint32_t k11, uint32_t k20, uint32_t k21, const float* obs_in,
                                 float* obs_out, hipStream_t s)
{
    const dim3 grid((p.N + BLOCK - 1) / BLOCK), blk(BLOCK);
    hipLaunchKernelGGL((reset_done_kernel<R, BLOCK, PMAX>), grid, blk, step_lds_bytes(p, BLOCK), s, p,
                       nobj_total, k10, k11, k20, k21, b.pool.layout_size, b.pool.cand_of, b.pool.cand_xy, b.dyn, b.obj,
                       obs_in, obs_out);
}


template <class R>
static void launch_group_r(const Params& p, const RolloutArgs& r, const DevBuffers& b, hipStream_t s)
{
    const dim3 grid((p.N + 3) / 4), blk(64);
    const PolicyArgs nopol = {};
#define GX_GROUP_LAUNCH(OPL, BPL, DEF)                                                                      \
    do {                                                                                                    \
        if (r.qacc)                                                                                         \
            hipLaunchKernelGGL((group_rollout_kernel<R, OPL, BPL, true, DEF, 0>), grid, blk, 0, s, p, r, nopol, b.dyn, b.obj, b.hist); \
        else                                                                                                \
            hipLaunchKernelGGL((group_rollout_kernel<R, OPL, BPL, false, DEF, 0>), grid, blk, 0, s, p, r, nopol, b.dyn, b.obj, b.hist); \
    } while (0)
    if (is_default_layout<R>(p)) GX_GROUP_LAUNCH(1, 1, true);
    else if (p.nobj <= 16 && p.bins <= 16) GX_GROUP_LAUNCH(1, 1, false);
    else if (p.nobj <= 32 && p.bins <= 16) GX_GROUP_LAUNCH(2, 1, false); // e.g. 8 hazards + 8 pillars (BASELINE config 5)
    else GX_GROUP_LAUNCH(5, 4, false);
#undef GX_GROUP_LAUNCH
}


template <class R, int kPol>
static void launch_policy_rp(const Params& p, const RolloutArgs& r, const PolicyArgs& pol, const DevBuffers& b,
                             hipStream_t s)
{
    constexpr int BT = (kPol >= 2) ? 256 : 64;
    const dim3 grid((p.N + BT / 16 - 1) / (BT / 16)), blk(BT);
    const size_t lds = sizeof(float) * (size_t)policy_lds_floats(p.D, R::NA, kPol);
    auto launch = [&](auto kern) {
        if (lds > 64 * 1024) // more dynamic LDS than the default cap: raise it for this kernel
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds);
        hipLaunchKernelGGL(kern, grid, blk, lds, s, p, r, pol, b.dyn, b.obj, b.hist);
    };
    if (is_default_layout<R>(p)) launch(group_rollout_kernel<R, 1, 1, false, true, kPol>);
    else launch(group_rollout_kernel<R, 1, 1, false, false, kPol>);
}



template <class R, int BLOCK, int PMAX>
static void launch_thread_rollout_bp(const Params& p, const RolloutArgs& r, const DevBuffers& b, hipStream_t s)
{
    const dim3 grid((p.N + BLOCK - 1) / BLOCK), blk(BLOCK);
    const size_t lds = sizeof(float) * (size_t)BLOCK * r.obs_stride;
    if (PMAX == 5 && is_default_layout<R>(p))
        hipLaunchKernelGGL((thread_rollout_kernel<R, BLOCK, 5, true>), grid, blk, lds, s, p, r, b.dyn, b.obj, b.hist);
    else
        hipLaunchKernelGGL((thread_rollout_kernel<R, BLOCK, PMAX, false>), grid, blk, lds, s, p, r, b.dyn, b.obj, b.hist);
}

template <class R, int BLOCK, int PMAX>
static void launch_commit_bp(const Params& p, const DevBuffers& b, int nobj_total, int n_rows, hipStream_t s)
{
    hipLaunchKernelGGL((commit_pending_kernel<R, BLOCK, PMAX>), dim3((p.N + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, p,
                       nobj_total, n_rows, b.rd_j, b.pool.cand_xy, b.dyn, b.obj);
}

template <class R>
void RobotLaunch<R>::commit_pending(const Params& p, const DevBuffers& b, int nobj_total, int n_rows, hipStream_t s)
{
    GX_DISPATCH_P(R, launch_commit_bp, 64, p, b, nobj_total, n_rows, s);
}

template <class R>
void RobotLaunch<R>::fake_table(const Params& p, const Pool& pl, int nobj_total, int M, hipStream_t s)
{
    if constexpr (!R::kRestFixed) {
        constexpr int B = 64;
        hipLaunchKernelGGL((fake_table_kernel<R, B>), dim3((M + B - 1) / B), dim3(B), 0, s, p, nobj_total, pl.layout_size,
                           pl.cand_of, pl.cand_xy, pl.fake);
    }
}

template <class R>
hipError_t RobotLaunch<R>::split(const Params& p, const RolloutArgs& r, float* tape, float4* obj0, float* entry,
                                 const DevBuffers& b, hipStream_t s, hipEvent_t hold, int which, int lanes, int n_shards,
                                 long long shard_stride, long long out_stride)
{
    SplitArgs sa;
    sa.tape = tape; sa.obj0 = obj0; sa.entry = entry; sa.lanes = lanes;
    sa.shard_stride = shard_stride; sa.out_stride = out_stride;
    if constexpr (R::kRestFixed) {
        if (p.P <= 5) return launch_split_p<R, 5>(p, r, sa, b, s, hold, which, n_shards);
        if (p.P <= 9) return launch_split_p<R, 9>(p, r, sa, b, s, hold, which, n_shards);
        return launch_split_p<R, 33>(p, r, sa, b, s, hold, which, n_shards);
    } else { // Ant, Walker: the dynamics pass is the lane-group form of the step
        if (p.P <= 5) return launch_split_group_p<R, 5>(p, r, sa, b, s, hold, which, n_shards);
        if (p.P <= 9) return launch_split_group_p<R, 9>(p, r, sa, b, s, hold, which, n_shards);
        return launch_split_group_p<R, 33>(p, r, sa, b, s, hold, which, n_shards);
    }
}
template <class R>
int RobotLaunch<R>::split_width()
{
    return SplitTape<R>::kW;
}
template <class R>
int RobotLaunch<R>::split_entry_width()
{
    return SplitTape<R>::kE;
}

template <class R>
void RobotLaunch<R>::thread_rollout(const Params& p, const RolloutArgs& r, const DevBuffers& b, hipStream_t s)
{
    GX_DISPATCH_BP_R(R, launch_thread_rollout_bp, p, r, b, s);
}

template <class R>
void RobotLaunch<R>::step(const Params& p, const DevBuffers& b, const float* act, float* obs, float* rew,
                          float* cost, float* done, float* qacc, hipStream_t s)
{
    GX_DISPATCH_BP_R(R, launch_step_bp, p, b, act, obs, rew, cost, done, qacc, s);
}
template <class R>
void RobotLaunch<R>::reset_apply(const Params& p, const DevBuffers& b, int nobj_total, uint32_t k10, uint32_t k11,
                                 uint32_t k20, uint32_t k21, float* obs, int* host_ls, hipStream_t s)
{
    GX_DISPATCH_BP_R(R, launch_reset_apply_bp, p, b, nobj_total, k10, k11, k20, k21, obs, host_ls, s);
}
template <class R>
void RobotLaunch<R>::reset_done(const Params& p, const DevBuffers& b, int nobj_total, uint32_t k10, uint32_t k11,
                                uint32_t k20, uint32_t k21, const float* obs_in, float* obs_out, hipStream_t s)
{
    GX_DISPATCH_BP_R(R, launch_reset_done_bp, p, b, nobj_total, k10, k11, k20, k21, obs_in, obs_out, s);
}
template <class R>
void RobotLaunch<R>::group(const Params& p, const RolloutArgs& r, const DevBuffers& b, hipStream_t s)
{
    launch_group_r<R>(p, r, b, s);
}
template <class R>
void RobotLaunch<R>::policy(const Params& p, const RolloutArgs& r, const PolicyArgs& pol, const DevBuffers& b,
                            int impl, hipStream_t s)
{
    if constexpr (R::kRestFixed) { // widths 128, 192, 256 in one launch: the light robots (the Ant's / Walker's step needs the registers)
        if (impl == 3) { launch_policy_rp<R, 3>(p, r, pol, b, s); return; }
        if (impl == 192) { launch_policy_rp<R, 192>(p, r, pol, b, s); return; }
        if (impl == 256) { launch_policy_rp<R, 256>(p, r, pol, b, s); return; }
    }
    if (impl == 2) launch_policy_rp<R, 2>(p, r, pol, b, s);
    else launch_policy_rp<R, 1>(p, r, pol, b, s);
}

} // namespace gx

// Explicit instantiation in two parts (guardx_amd/build.py compiles them with different scheduling strategies): the
// two-kernel rollout -- whose dynamics pass is one wave per SIMD -- and everything else of a robot.
#define GX_INSTANTIATE_SPLIT(R)                                                                                          \
    template hipError_t RobotLaunch<R>::split(const Params&, const RolloutArgs&, float*, float4*, float*, const DevBuffers&, \
                                              hipStream_t, hipEvent_t, int, int, int, long long, long long);          \
    template int RobotLaunch<R>::split_width();                                                                          \
    template int RobotLaunch<R>::split_entry_width();
#define GX_INSTANTIATE_REST(R)                                                                                           \
    template void RobotLaunch<R>::step(const Params&, const DevBuffers&, const float*, float*, float*, float*, float*,   \
                                       float*, hipStream_t);                                                             \
    template void RobotLaunch<R>::reset_apply(const Params&, const DevBuffers&, int, uint32_t, uint32_t, uint32_t,       \
                                              uint32_t, float*, int*, hipStream_t);                                      \
    template void RobotLaunch<R>::reset_done(const Params&, const DevBuffers&, int, uint32_t, uint32_t, uint32_t,        \
                                             uint32_t, const float*, float*, hipStream_t);                               \
    template void RobotLaunch<R>::group(const Params&, const RolloutArgs&, const DevBuffers&, hipStream_t);              \
    template void RobotLaunch<R>::thread_rollout(const Params&, const RolloutArgs&, const DevBuffers&, hipStream_t);     \
    template void RobotLaunch<R>::policy(const Params&, const RolloutArgs&, const PolicyArgs&, const DevBuffers&, int,   \
                                         hipStream_t);                                                                   \
    template void RobotLaunch<R>::commit_pending(const Params&, const DevBuffers&, int, int, hipStream_t);               \
    template void RobotLaunch<R>::fake_table(const Params&, const Pool&, int, int, hipStream_t);

