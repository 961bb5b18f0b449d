// gx_kernels.hip -- HIP kernels of the GUARD batched environment step (gfx950).
//
// One thread owns one environment.  Environment state is struct-of-arrays of
// float4 (`dyn`, `obj`), so each wave64 load/store is one coalesced 1 KiB
// transaction.  The learner-facing observation is env-major (N, D) (the learner
// writes obs_buf[:, t, :] = obs, reference trpo.py:58), so every thread builds
// its D-float row in LDS -- the lidar bins are scatter-max'ed in place there --
// and the block then streams the whole tile out as contiguous float4 stores.
//
// Reference lines: /root/reference/safe_rl_envs/safe_rl_envs/envs/engine.py.
#include "gx_kernels.h"

namespace gx {

// ---------------------------------------------------------------------------
// observation row (engine.py:738-778) built in this thread's LDS row.
// `ob` holds the object pairs: ob[k] = (obj 2k xy, obj 2k+1 xy); obj 0 = goal.
// ---------------------------------------------------------------------------
template <int PMAX>
GX_D bool build_obs_row(const Params& p, float* row, const float (&pose)[4],
                        const float4 (&ob)[PMAX], float cx, float cy, float ct,
                        const PtState& st, float vel0, float vel1, float acc0, float acc1)
{
    bool bad = false;
    if (p.off_acc >= 0) {
        row[p.off_acc] = acc0; row[p.off_acc + 1] = acc1;
        bad = bad || notfinite(acc0) || notfinite(acc1);
    }
    if (p.off_ctrl >= 0) {
        row[p.off_ctrl] = cx; row[p.off_ctrl + 1] = cy; row[p.off_ctrl + 2] = ct;
        bad = bad || notfinite(cx) || notfinite(cy) || notfinite(ct);
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
    if (p.off_hl >= 0) {
        float* r = row + p.off_hl;
        for (int b = 0; b < p.bins; ++b) r[b] = 0.0f;
#pragma unroll
        for (int k = 0; k < PMAX; ++k) {
            // objects 2k and 2k+1; object 0 is the goal
            if (k > 0 && 2 * k < p.nobj) bad = lidar_one(p, r, ob[k].x, ob[k].y, pose) || bad;
            if (2 * k + 1 < p.nobj) bad = lidar_one(p, r, ob[k].z, ob[k].w, pose) || bad;
        }
    }
    if (p.off_qpos >= 0) {
        row[p.off_qpos] = st.x; row[p.off_qpos + 1] = st.y; row[p.off_qpos + 2] = st.th;
        bad = bad || notfinite(st.x) || notfinite(st.y) || notfinite(st.th);
    }
    if (p.off_qvel >= 0) {
        row[p.off_qvel] = st.vx; row[p.off_qvel + 1] = st.vy; row[p.off_qvel + 2] = st.om;
        bad = bad || notfinite(st.vx) || notfinite(st.vy) || notfinite(st.om);
    }
    if (p.off_vel >= 0) {
        row[p.off_vel] = vel0; row[p.off_vel + 1] = vel1;
        bad = bad || notfinite(vel0) || notfinite(vel1);
    }
    return bad;
}

// stream the block's LDS tile (nenv rows of D floats, env-major) to global
template <int BLOCK>
GX_D void flush_tile(const float* tile, float* gbase, int total)
{
    const int nvec = total >> 2;
    const float4* t4 = reinterpret_cast<const float4*>(tile);
    float4* g4 = reinterpret_cast<float4*>(gbase);
    for (int v = threadIdx.x; v < nvec; v += BLOCK) g4[v] = t4[v];
    for (int k = (nvec << 2) + threadIdx.x; k < total; k += BLOCK) gbase[k] = tile[k];
}

GX_D float dist2(float ax, float ay, float bx, float by)
{
    const float dx = ax - bx, dy = ay - by;
    return sqrtf(dx * dx + dy * dy);
}

// ---------------------------------------------------------------------------
// Engine.step (engine.py:469-495 + mjx_step :659-700), Point robot.
// ---------------------------------------------------------------------------
template <int BLOCK, int PMAX, bool kQacc>
__global__ __launch_bounds__(BLOCK) void step_kernel(Params p, const float2* __restrict__ act,
                                                     float4* __restrict__ dyn,
                                                     const float4* __restrict__ obj,
                                                     float4* __restrict__ hist,
                                                     float* __restrict__ obs,
                                                     float* __restrict__ rew,
                                                     float* __restrict__ cost,
                                                     float* __restrict__ done,
                                                     float* __restrict__ qacc_out)
{
    extern __shared__ float4 tile4[];
    float* tile = reinterpret_cast<float*>(tile4);
    const int tid = threadIdx.x;
    const int env0 = blockIdx.x * BLOCK;
    const int i = env0 + tid;
    const bool live = i < p.N;

    // ---- coalesced loads (arrays are padded to Npad, every lane may load)
    const float2 a = live ? act[i] : make_float2(0.f, 0.f);
    const float4 d0 = dyn[i];
    const float4 d1 = dyn[p.Npad + i];
    const float4 d2 = dyn[2 * p.Npad + i];
    float4 ob[PMAX];
#pragma unroll
    for (int k = 0; k < PMAX; ++k)
        ob[k] = (k < p.P) ? obj[(size_t)k * p.Npad + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 hs = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.hist_on) hs = hist[i];

    PtState st = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y};
    const float P1x = d1.z, P1y = d1.w; // last_data.xpos
    const float pc = d2.x, ps = d2.y;   // pre-step xmat (heading)
    const float last_done = d2.z;       // _last_done after update_data
    const float steps = d2.w;

    // convert_action :672-685
    const float cx = pc * a.x, cy = ps * a.x, ct = a.y;

    float pose[4], qacc[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < p.physics_steps; ++k) point_substep<kQacc>(st, cx, cy, ct, pose, qacc);

    // ego_vel_acc :902-929
    float vel0 = 0.f, vel1 = 0.f, acc0 = 0.f, acc1 = 0.f;
    if (p.hist_on) {
        float plx = pose[0], ply = pose[1], pllx = pose[0], plly = pose[1];
        if (p.have_last) {
            if (!(last_done > 0.0f)) { plx = P1x; ply = P1y; }
            if (p.have_last_last) {
                if (hs.z + last_done > 0.0f) { pllx = plx; plly = ply; }
                else { pllx = hs.x; plly = hs.y; }
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

    float* row = tile + tid * p.D;
    const bool bad = build_obs_row<PMAX>(p, row, pose, ob, cx, cy, ct, st, vel0, vel1, acc0, acc1);

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
        if (k > 0 && 2 * k < p.nobj) {
            const float dh = dist2(ob[k].x, ob[k].y, pose[0], pose[1]);
            float below = dh < p.hazards_size ? dh : p.hazards_size;
            if (dh != dh) below = dh;
            cs = cs + (p.hazards_size - below);
        }
        if (2 * k + 1 < p.nobj) {
            const float dh = dist2(ob[k].z, ob[k].w, pose[0], pose[1]);
            float below = dh < p.hazards_size ? dh : p.hazards_size;
            if (dh != dh) below = dh;
            cs = cs + (p.hazards_size - below);
        }
    }

    // NaN/Inf guard :696-699, timeout + step counter :492-493
    if (bad) { r = 0.0f; dn = 1.0f; }
    if (steps > p.num_steps_f) dn = 1.0f;
    const float nsteps = dn > 0.0f ? 0.0f : steps + 1.0f;

    if (live) {
        dyn[i] = make_float4(st.x, st.y, st.th, st.vx);
        dyn[p.Npad + i] = make_float4(st.vy, st.om, pose[0], pose[1]);
        dyn[2 * p.Npad + i] = make_float4(pose[2], pose[3], dn, nsteps);
        if (p.hist_on) hist[i] = make_float4(P1x, P1y, last_done, 0.f);
        rew[i] = r;
        cost[i] = cs;
        done[i] = dn;
        if (kQacc) {
            qacc_out[3 * i] = qacc[0];
            qacc_out[3 * i + 1] = qacc[1];
            qacc_out[3 * i + 2] = qacc[2];
        }
    }

    __syncthreads();
    const int nenv = min(BLOCK, p.N - env0);
    flush_tile<BLOCK>(tile, obs + (size_t)env0 * p.D, nenv * p.D);
}

// ---------------------------------------------------------------------------
// layout rejection sampler: sample_layout (engine.py:546-572) for candidate j,
// key_j = split(key, M)[j] (:263).  Placed objects live in LDS, object-major.
// ---------------------------------------------------------------------------
constexpr int kSampleBlock = 256;

__global__ __launch_bounds__(kSampleBlock) void sample_kernel(SampleParams sp,
                                                              uint8_t* __restrict__ ok,
                                                              float2* __restrict__ cand_xy,
                                                              int* __restrict__ wave_cnt)
{
    extern __shared__ float4 smem4[];
    float2* placed = reinterpret_cast<float2*>(smem4); // [nobj_total][kSampleBlock]
    const int tid = threadIdx.x;
    const int j = blockIdx.x * kSampleBlock + tid;
    const bool live = j < sp.M;
    uint32_t r0, r1;
    split_at(sp.k0, sp.k1, (uint32_t)sp.M, (uint32_t)(live ? j : 0), r0, r1);
    bool success = true;
    const int nobj = sp.nobj_total;
    for (int o = 0; o < nobj; ++o) {
        const int ty = (o == 0) ? 0 : (o == nobj - 1 ? 2 : 1);
        const float lox = sp.lo_x[ty], hix = sp.hi_x[ty], loy = sp.lo_y[ty], hiy = sp.hi_y[ty];
        bool conflicted = true;
        float px = -__builtin_inff(), py = -__builtin_inff();
        for (int t = 0; t < 10; ++t) {
            uint32_t n0, n1, g0, g1, u0, u1, v0, v1;
            split2(r0, r1, n0, n1, g0, g1); // rng, rng1 = split(rng)
            r0 = n0; r1 = n1;
            split2(g0, g1, u0, u1, v0, v1); // draw_placement :618
            const float cx = uniform_f(u0, u1, lox, hix);
            const float cy = uniform_f(v0, v1, loy, hiy);
            bool flag = true;
            for (int q = 0; q < o; ++q) {
                const float2 pq = placed[q * kSampleBlock + tid];
                const float dist = dist2(cx, cy, pq.x, pq.y);
                const int tq = (q == 0) ? 0 : 1;
                if (dist < sp.thr[tq][ty]) flag = false;
            }
            if (flag) { px = cx; py = cy; conflicted = false; }
        }
        placed[o * kSampleBlock + tid] = make_float2(px, py);
        if (conflicted) success = false;
    }
    {
        const float2 g = placed[tid], rb = placed[(nobj - 1) * kSampleBlock + tid];
        const float d = dist2(rb.x, rb.y, g.x, g.y);
        if (d < sp.min_rg) success = false;
    }
    success = success && live;
    if (live) ok[j] = success ? 1 : 0;
    if (success)
        for (int o = 0; o < nobj; ++o) cand_xy[(size_t)j * nobj + o] = placed[o * kSampleBlock + tid];
    const unsigned long long m = __ballot(success);
    if ((tid & 63) == 0 && live) wave_cnt[j >> 6] = __popcll(m);
}

// exclusive scan of the per-wave valid counts (one block)
constexpr int kScanBlock = 1024;
__global__ __launch_bounds__(kScanBlock) void scan_kernel(const int* __restrict__ cnt,
                                                          int* __restrict__ off, int W,
                                                          int* __restrict__ total)
{
    __shared__ int part[kScanBlock];
    const int tid = threadIdx.x;
    const int chunk = (W + kScanBlock - 1) / kScanBlock;
    const int beg = tid * chunk, end = min(W, beg + chunk);
    int s = 0;
    for (int k = beg; k < end; ++k) s += cnt[k];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < kScanBlock; d <<= 1) {
        const int v = (tid >= d) ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s; // exclusive prefix of this chunk
    for (int k = beg; k < end; ++k) { off[k] = run; run += cnt[k]; }
    if (tid == kScanBlock - 1) *total = part[tid];
}

// idx = where(success > 0)[0]  (engine.py:436): ordered compaction
__global__ __launch_bounds__(kSampleBlock) void compact_kernel(int M, const uint8_t* __restrict__ ok,
                                                               const int* __restrict__ off,
                                                               int* __restrict__ cand_of)
{
    const int j = blockIdx.x * kSampleBlock + threadIdx.x;
    const bool v = (j < M) && ok[j];
    const unsigned long long m = __ballot(v);
    const int lane = threadIdx.x & 63;
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (v) cand_of[off[j >> 6] + rank] = j;
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

template <int BLOCK, int PMAX>
__global__ __launch_bounds__(BLOCK) void reset_apply_kernel(Params p, int nobj_total, uint32_t k10,
                                                            uint32_t k11, uint32_t k20, uint32_t k21,
                                                            const int* __restrict__ layout_size,
                                                            const int* __restrict__ cand_of,
                                                            const float2* __restrict__ cand_xy,
                                                            float4* __restrict__ dyn,
                                                            float4* __restrict__ obj,
                                                            float* __restrict__ obs)
{
    extern __shared__ float4 tile4[];
    float* tile = reinterpret_cast<float*>(tile4);
    const int L = *layout_size;
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
    const PtState st = {rx, ry, 0.f, 0.f, 0.f, 0.f};
    const float pose[4] = {rx, ry, 1.0f, 0.0f};
    float* row = tile + tid * p.D;
    build_obs_row<PMAX>(p, row, pose, ob, 0.f, 0.f, 0.f, st, 0.f, 0.f, 0.f, 0.f);
    if (live) {
        const float4 d2 = dyn[2 * p.Npad + i];
        dyn[i] = make_float4(rx, ry, 0.f, 0.f);
        dyn[p.Npad + i] = make_float4(0.f, 0.f, rx, ry);
        dyn[2 * p.Npad + i] = make_float4(1.0f, 0.0f, d2.z, 0.0f); // _done kept, _steps = 0 (:463)
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
template <int BLOCK, int PMAX>
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
    const float4 d2 = dyn[2 * p.Npad + i];
    const bool dn = live && (d2.z > 0.0f) && (L > 0);
    const int any = __syncthreads_or(dn ? 1 : 0);
    const int nenv = min(BLOCK, p.N - env0);
    const int total = nenv * p.D;
    if (!any && obs_in == obs_out) return; // nothing to do for this tile
    // stage the old rows (self._obs) in LDS
    {
        const int nvec = total >> 2;
        const float4* g4 = reinterpret_cast<const float4*>(obs_in + (size_t)env0 * p.D);
        for (int v = tid; v < nvec; v += BLOCK) tile4[v] = g4[v];
        const float* g = obs_in + (size_t)env0 * p.D;
        for (int k = (nvec << 2) + tid; k < total; k += BLOCK) tile[k] = g[k];
    }
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
        const PtState st = {rx, ry, 0.f, 0.f, 0.f, 0.f};
        const float pose[4] = {rx, ry, 1.0f, 0.0f};
        build_obs_row<PMAX>(p, tile + tid * p.D, pose, ob, 0.f, 0.f, 0.f, st, 0.f, 0.f, 0.f, 0.f);
        const float4 d1 = dyn[p.Npad + i];
        dyn[i] = make_float4(rx, ry, 0.f, 0.f);
        dyn[p.Npad + i] = make_float4(0.f, 0.f, d1.z, d1.w);
#pragma unroll
        for (int k = 0; k < PMAX; ++k)
            if (k < p.P) obj[(size_t)k * p.Npad + i] = ob[k];
    }
    __syncthreads();
    flush_tile<BLOCK>(tile, obs_out + (size_t)env0 * p.D, total);
}

// ---------------------------------------------------------------------------
// probes
// ---------------------------------------------------------------------------
__global__ void math_probe_kernel(int n, const float* x, const float* y, float* s, float* c,
                                  float* at2, float* ex)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float ss, cc;
    sincos_f(x[i], ss, cc);
    s[i] = ss; c[i] = cc;
    at2[i] = atan2_f(y[i], x[i]);
    ex[i] = exp_f(x[i]);
}

__global__ void split_probe_kernel(uint32_t k0, uint32_t k1, int n, uint32_t* out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t a, b;
    split_at(k0, k1, (uint32_t)n, (uint32_t)j, a, b);
    out[2 * j] = a; out[2 * j + 1] = b;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
size_t step_lds_bytes(const Params& p, int block) { return (size_t)block * p.D * sizeof(float); }

int pick_block(const Params& p)
{
    // small batches: one wave per workgroup spreads the envs over more CUs (latency regime);
    // large batches: 256-thread workgroups when the obs tile fits comfortably in LDS
    if (p.N <= 32768) return 64;
    return (step_lds_bytes(p, 256) <= 48 * 1024) ? 256 : 64;
}

template <int BLOCK, int PMAX>
static void launch_step_bp(const Params& p, const DevBuffers& b, const float* act, float* obs,
                           float* rew, float* cost, float* done, float* qacc, hipStream_t s)
{
    const dim3 grid((p.N + BLOCK - 1) / BLOCK), blk(BLOCK);
    const size_t lds = step_lds_bytes(p, BLOCK);
    if (qacc)
        hipLaunchKernelGGL((step_kernel<BLOCK, PMAX, true>), grid, blk, lds, s, p,
                           reinterpret_cast<const float2*>(act), b.dyn, b.obj, b.hist, obs, rew, cost,
                           done, qacc);
    else
        hipLaunchKernelGGL((step_kernel<BLOCK, PMAX, false>), grid, blk, lds, s, p,
                           reinterpret_cast<const float2*>(act), b.dyn, b.obj, b.hist, obs, rew, cost,
                           done, qacc);
}

#define GX_DISPATCH_BP(FN, ...)                                        \
    do {                                                               \
        const int blk_ = pick_block(p);                                \
        if (p.P <= 5) {                                                \
            if (blk_ == 64) FN<64, 5>(__VA_ARGS__);                    \
            else FN<256, 5>(__VA_ARGS__);                              \
        } else if (p.P <= 9) {                                         \
            if (blk_ == 64) FN<64, 9>(__VA_ARGS__);                    \
            else FN<256, 9>(__VA_ARGS__);                              \
        } else {                                                       \
            if (blk_ == 64) FN<64, 33>(__VA_ARGS__);                   \
            else FN<256, 33>(__VA_ARGS__);                             \
        }                                                              \
    } while (0)

void launch_step(const Params& p, const DevBuffers& b, const float* act, float* obs, float* rew,
                 float* cost, float* done, float* qacc, hipStream_t s)
{
    GX_DISPATCH_BP(launch_step_bp, p, b, act, obs, rew, cost, done, qacc, s);
}

void launch_sample(const SampleParams& sp, const DevBuffers& b, hipStream_t s)
{
    const int M = sp.M, W = (M + 63) / 64;
    const int grid = (M + kSampleBlock - 1) / kSampleBlock;
    const size_t lds = (size_t)sp.nobj_total * kSampleBlock * sizeof(float2);
    hipLaunchKernelGGL(sample_kernel, dim3(grid), dim3(kSampleBlock), lds, s, sp, b.cand_ok, b.cand_xy,
                       b.wave_cnt);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(kScanBlock), 0, s, b.wave_cnt, b.wave_off, W,
                       b.layout_size);
    hipLaunchKernelGGL(compact_kernel, dim3(grid), dim3(kSampleBlock), 0, s, M, b.cand_ok, b.wave_off,
                       b.cand_of);
}

template <int BLOCK, int PMAX>
static void launch_reset_apply_bp(const Params& p, const DevBuffers& b, int nobj_total, uint32_t k10,
                                  uint32_t k11, uint32_t k20, uint32_t k21, float* obs, hipStream_t s)
{
    const dim3 grid((p.N + BLOCK - 1) / BLOCK), blk(BLOCK);
    hipLaunchKernelGGL((reset_apply_kernel<BLOCK, PMAX>), grid, blk, step_lds_bytes(p, BLOCK), s, p,
                       nobj_total, k10, k11, k20, k21, b.layout_size, b.cand_of, b.cand_xy, b.dyn, b.obj,
                       obs);
}

void launch_reset_apply(const Params& p, const DevBuffers& b, int nobj_total, uint32_t k10,
                        uint32_t k11, uint32_t k20, uint32_t k21, float* obs, hipStream_t s)
{
    GX_DISPATCH_BP(launch_reset_apply_bp, p, b, nobj_total, k10, k11, k20, k21, obs, s);
}

template <int BLOCK, int PMAX>
static void launch_reset_done_bp(const Params& p, const DevBuffers& b, int nobj_total, uint32_t k10,
                                 uint32_t k11, uint32_t k20, uint32_t k21, const float* obs_in,
                                 float* obs_out, hipStream_t s)
{
    const dim3 grid((p.N + BLOCK - 1) / BLOCK), blk(BLOCK);
    hipLaunchKernelGGL((reset_done_kernel<BLOCK, PMAX>), grid, blk, step_lds_bytes(p, BLOCK), s, p,
                       nobj_total, k10, k11, k20, k21, b.layout_size, b.cand_of, b.cand_xy, b.dyn, b.obj,
                       obs_in, obs_out);
}

void launch_reset_done(const Params& p, const DevBuffers& b, int nobj_total, uint32_t k10,
                       uint32_t k11, uint32_t k20, uint32_t k21, const float* obs_in, float* obs_out,
                       hipStream_t s)
{
    GX_DISPATCH_BP(launch_reset_done_bp, p, b, nobj_total, k10, k11, k20, k21, obs_in, obs_out, s);
}

void launch_math_probe(int n, const float* x, const float* y, float* s_, float* c, float* at2,
                       float* ex, hipStream_t s)
{
    hipLaunchKernelGGL(math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, x, y, s_, c, at2, ex);
}

void launch_split_probe(uint32_t k0, uint32_t k1, int n, uint32_t* out, hipStream_t s)
{
    hipLaunchKernelGGL(split_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, s, k0, k1, n, out);
}

} // namespace gx
