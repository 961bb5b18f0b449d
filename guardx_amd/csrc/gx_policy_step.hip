// gx_policy_step.hip -- `ac.step(o)` of MLPActorCritic(hidden_sizes=(H,H), tanh) for the hidden widths the learner's
// command line offers beyond its default (safe_rl_libX/trpo/trpo.py:606-607 --hid/--l; trpo_core.py:110-173): H = 128,
// 256 (any multiple of 64 up to 256; 64 too, as a cross-check of the fused kernel).
//
// The fused closed-loop kernel (gx_policy.h) keeps both networks in LDS: 56 KB at H = 64, but 179 KB at H = 128 and
// 616 KB at H = 256 -- they do not fit.  For those widths gx_rollout_policy alternates two launches per control step:
// this kernel (policy over all envs: action, mu, logp, value) and the ordinary fused step + reset_done launch.  Same
// arithmetic as the fused kernel and the CPU checker (oracle/gx_oracle.c:mlp_forward): every hidden unit is ONE
// sequential fmaf chain over its inputs, the output layer 16 partial sums (partial l: units 64 c + 4 l + j) folded by a
// butterfly, tanh / exp / log / sincos the shared polynomials -- bit-identical results.
//
// Work split: a workgroup of 2 H / 64 waves serves kEnv = 8 envs; wave w evaluates 64 hidden units (chunk w % (H/64))
// of network w / (H/64) (0 = actor, 1 = critic) for all 8 envs at once -- one coalesced weight load ([in][out]
// transposed copy in global memory, L2 resident: every workgroup streams the same 0.7 MB at H = 256) feeds 8 fmaf
// chains; inputs are LDS broadcasts ([k][env] layout: two 16-byte reads give the 8 envs' values).  This is HBM-trivial,
// VALU work: 2 (D + H) H fmaf per env-step (154 k at H = 256).
#include "gx_kernels.h"
#include "gx_policy.h"

namespace gx {

constexpr int kPsEnv = 8; // envs per workgroup

// Wt1[k][H] = W1[j][k] (k < D; rows D .. pad4(D)-1 are zero: the MFMA form consumes K in fours), Wt2[k][H] = W2[j][k],
// for both networks: [pi Wt1 | pi Wt2 | v Wt1 | v Wt2]
__global__ void policy_transpose_kernel(const float* __restrict__ params, float* __restrict__ wt, int D, int A, int H)
{
    const int Dp = pad4(D);
    const int per = Dp * H + H * H;
    const int n = 2 * per;
    const int msz_pi = H * D + H + H * H + H + A * H + A;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int net = i / per, r = i - net * per;
        const float* g = params + (net ? msz_pi : 0);
        if (r < Dp * H) {
            const int k = r / H, j = r - k * H;
            wt[i] = k < D ? g[j * D + k] : 0.0f;
        } else {
            const int r2 = r - Dp * H, k = r2 / H, j = r2 - k * H;
            wt[i] = g[H * D + H + j * H + k];
        }
    }
}

struct PolicyStepArgs {
    const float* params;  // torch layout (include/guardx.h)
    const float* wt;      // transposed hidden-layer weights (policy_transpose_kernel)
    const float* obs;     // [N][D] the observation the policy sees
    uint32_t seed0, seed1, tnoise; // noise key, policy step counter of THIS step (t0 + t)
    int N, D, A, env_offset;
    int mode;             // 0: ac.step -> obs_in, act, mu, logp, val (one time step);  1: critic only -> val_last, obs_last
    float* obs_in;        // [N][D] row block of step t
    float* act;           // [N][A]
    float* mu;            // [N][A]
    float* logp;          // [N]
    float* val;           // [N]  (mode 1: val_last)
    float* obs_last;      // mode 1: [N][D]
    float* logstd;        // [A], written by workgroup 0 in mode 0
};

// per env: the noise, the action, log pi(a | o), and the outputs of this step (ac.step, trpo_core.py:166-173)
template <int E>
GX_D void policy_step_tail(const PolicyStepArgs& a, const float* outs, int env0, int tid, int H, int msz_pi)
{
    const int D = a.D, A = a.A;
    if (tid < E) {
        const int e = tid, env = env0 + e;
        if (env < a.N) {
            const float v = outs[e * (A + 1) + A];
            a.val[env] = v;
            if (a.mode == 0) {
                const float* gls = a.params + msz_pi + (H * D + H + H * H + H + H + 1);
                float lp = 0.0f;
                for (int pr = 0; 2 * pr < A; ++pr) { // one counter per pair of action dimensions (trpo_core.py:166-173)
                    float z[2];
                    normal_pair(a.seed0, a.seed1, (uint32_t)(a.env_offset + env), a.tnoise * 16u + (uint32_t)pr, z[0], z[1]);
                    for (int q = 0; q < 2; ++q) {
                        const int d = 2 * pr + q;
                        const float sd = exp_f(gls[d]);           // std = exp(log_std)   trpo_core.py:123
                        const float lsd = log_f(sd);              // torch.log(pi.stddev) trpo_core.py:173
                        const float m = outs[e * (A + 1) + d];
                        const float act = fmaf(sd, z[q], m);
                        const float df = act - m;
                        const float var = sd * sd;
                        lp = lp + ((-(df * df) / (2.0f * var) - lsd) - 0.9189385332046727f);
                        a.act[(size_t)env * A + d] = act;
                        a.mu[(size_t)env * A + d] = m;
                    }
                }
                a.logp[env] = lp;
            }
        }
    }
    if (a.mode == 0 && blockIdx.x == 0 && tid < A) {
        const float* gls = a.params + msz_pi + (H * D + H + H * H + H + H + 1);
        a.logstd[tid] = log_f(exp_f(gls[tid]));
    }
}

// acc[e] = fmaf(x[k][e], w[k], acc[e]) for k = 0 .. K-1 (ascending: one sequential chain per env), the weights of unit
// `u` fetched kWB at a time and one block AHEAD of the arithmetic: a plain loop waits one L2 round trip per k (the
// compiler does not hoist the loads of a run-time-length loop): 172 round trips per launch at H = 128.
constexpr int kWB = 16;
template <int E>
GX_D void chain_layer(float (&acc)[E], const float* __restrict__ wt, int H, int u, const float* xk, int K)
{
    static_assert(E == 8, "two float4 reads per k");
    float w[kWB], wn[kWB];
#pragma unroll
    for (int i = 0; i < kWB; ++i) w[i] = i < K ? wt[(size_t)i * H + u] : 0.0f;
#pragma unroll 1 // (K = H is a compile-time constant at the second layer: fully unrolled, every load of the layer is
                 // hoisted to the top and the kernel spills 9 KB per thread)
    for (int k0 = 0; k0 < K; k0 += kWB) {
#pragma unroll
        for (int i = 0; i < kWB; ++i) wn[i] = (k0 + kWB + i) < K ? wt[(size_t)(k0 + kWB + i) * H + u] : 0.0f;
#pragma unroll
        for (int i = 0; i < kWB; ++i) {
            if (k0 + i < K) {
                const float4 xa = *reinterpret_cast<const float4*>(xk + (k0 + i) * E);
                const float4 xb = *reinterpret_cast<const float4*>(xk + (k0 + i) * E + 4);
                const float ww = w[i];
                acc[0] = fmaf(xa.x, ww, acc[0]); acc[1] = fmaf(xa.y, ww, acc[1]); acc[2] = fmaf(xa.z, ww, acc[2]); acc[3] = fmaf(xa.w, ww, acc[3]);
                acc[4] = fmaf(xb.x, ww, acc[4]); acc[5] = fmaf(xb.y, ww, acc[5]); acc[6] = fmaf(xb.z, ww, acc[6]); acc[7] = fmaf(xb.w, ww, acc[7]);
            }
        }
#pragma unroll
        for (int i = 0; i < kWB; ++i) w[i] = wn[i];
    }
}

template <int H>
__global__ __launch_bounds__(2 * H) void policy_step_kernel(PolicyStepArgs a)
{
    constexpr int U = H / 64, E = kPsEnv, NT = 2 * H;
    extern __shared__ float4 ps_lds4[];
    float* lds = reinterpret_cast<float*>(ps_lds4);
    const int D = a.D, A = a.A;
    float* xs = lds;                          // [D][E]
    float* h1 = xs + pad4(D) * E;             // [2][H][E]
    float* h2 = h1 + 2 * H * E;               // [2][H][E]
    float* outs = h2 + 2 * H * E;             // [E][A + 1]: mu.., value
    const int tid = threadIdx.x, wave = tid >> 6, j = tid & 63;
    const int net = wave / U, u = 64 * (wave % U) + j;
    const int env0 = blockIdx.x * E;
    const int msz_pi = H * D + H + H * H + H + A * H + A;
    const float* g = a.params + (net ? msz_pi : 0);
    const float *b1 = g + H * D, *b2 = b1 + H + H * H;
    const float* wt1 = a.wt + (size_t)net * (pad4(D) * H + H * H);
    const float* wt2 = wt1 + pad4(D) * H;

    for (int i = tid; i < D * E; i += NT) {
        const int e = i / D, k = i - e * D;
        const int env = env0 + e;
        const float x = env < a.N ? a.obs[(size_t)env * D + k] : 0.0f;
        xs[k * E + e] = x;
        if (env < a.N) {
            if (a.mode == 0) a.obs_in[(size_t)env * D + k] = x;
            else a.obs_last[(size_t)env * D + k] = x;
        }
    }
    __syncthreads();
    const bool skip = a.mode == 1 && net == 0; // the bootstrap value needs the critic only
    float acc[E];
    if (!skip) {
        const float bb = b1[u];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = bb;
        chain_layer<E>(acc, wt1, H, u, xs, D);
        float* hp = h1 + ((size_t)net * H + u) * E;
        *reinterpret_cast<float4*>(hp) = make_float4(tanh_f(acc[0]), tanh_f(acc[1]), tanh_f(acc[2]), tanh_f(acc[3]));
        *reinterpret_cast<float4*>(hp + 4) = make_float4(tanh_f(acc[4]), tanh_f(acc[5]), tanh_f(acc[6]), tanh_f(acc[7]));
    }
    __syncthreads();
    if (!skip) {
        const float bb = b2[u];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = bb;
        const float* hin = h1 + (size_t)net * H * E;
        chain_layer<E>(acc, wt2, H, u, hin, H);
        float* hp = h2 + ((size_t)net * H + u) * E;
        *reinterpret_cast<float4*>(hp) = make_float4(tanh_f(acc[0]), tanh_f(acc[1]), tanh_f(acc[2]), tanh_f(acc[3]));
        *reinterpret_cast<float4*>(hp + 4) = make_float4(tanh_f(acc[4]), tanh_f(acc[5]), tanh_f(acc[6]), tanh_f(acc[7]));
    }
    __syncthreads();
    // output layers: task (env e, output o) on 16 lanes; o < A: mu_o (actor), o == A: the value (critic)
    const int l = tid & 15;
    for (int task = tid >> 4; task < E * (A + 1); task += NT / 16) {
        const int e = task / (A + 1), o = task - e * (A + 1);
        const int nt = o == A ? 1 : 0, oo = nt ? 0 : o;
        if (a.mode == 1 && !nt) continue;                   // (16-lane groups take the branch together)
        const float* gg = a.params + (nt ? msz_pi : 0);
        const float* W3 = gg + H * D + H + H * H + H;
        const float* b3 = W3 + (nt ? 1 : A) * H;
        const float* hh = h2 + (size_t)nt * H * E;
        float pp = 0.0f;
#pragma unroll
        for (int c = 0; c < U; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int unit = 64 * c + 4 * l + q;
                pp = fmaf(hh[unit * E + e], W3[oo * H + unit], pp);
            }
        pp = pp + __shfl_xor(pp, 8, 16);
        pp = pp + __shfl_xor(pp, 4, 16);
        pp = pp + __shfl_xor(pp, 2, 16);
        pp = pp + __shfl_xor(pp, 1, 16);
        if (l == 0) outs[e * (A + 1) + o] = b3[oo] + pp;
    }
    __syncthreads();
    policy_step_tail<E>(a, outs, env0, tid, H, msz_pi);
}

// ---------------------------------------------------------------------------------------------------------------
// The same step with the two hidden layers on the matrix cores (the default for the wide networks).
// v_mfma_f32_16x16x4_f32 accumulates exactly like a sequential fmaf chain over k (tools/probes/mfma_f32_probe.hip,
// gx_policy.h), so the results are those of the VALU form above and of the checker, bit for bit -- with 16x fewer
// instructions on the chain that bounded it (172 / 300 dependent k-steps of eight fmaf each).
// A workgroup of 8 waves serves 16 envs: wave w works on network w / 4 (0 actor, 1 critic) and on the H / 64 unit tiles
// (16 units each) [ (w % 4) H / 64, .. ): H1[16 envs][16 units] += X[16][K] Wt[K][16], K in steps of 4.  Operands: A lane
// = k * 16 + env from LDS (the observation rows, then the first hidden layer), B lane = k * 16 + unit straight from the
// transposed weights in global memory (16 consecutive floats per k: coalesced, L2 resident), D lane holds envs
// 4 (lane / 16) .. + 3 of unit lane % 16.  The next k-step's operands are fetched before this one's MFMAs issue.
// ---------------------------------------------------------------------------------------------------------------
typedef float ps_f4 __attribute__((ext_vector_type(4)));
constexpr int kPmEnv = 16;

constexpr int kMB = 8; // k-steps (of 4) whose operands are in flight together
template <int TT>
GX_D void mfma_fetch(float (&av)[kMB], float (&bv)[kMB][TT], const float* ap, const float* bp, int H, int s0, int ns)
{
#pragma unroll
    for (int i = 0; i < kMB; ++i) {
        const int sidx = s0 + i;
        if (sidx < ns) {                         // wave-uniform
            av[i] = ap[4 * sidx];
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) bv[i][tt] = bp[(size_t)(4 * sidx) * H + 16 * tt];
        }
    }
}
template <int TT>
GX_D void mfma_issue(ps_f4 (&acc)[TT], const float (&av)[kMB], const float (&bv)[kMB][TT], int s0, int ns)
{
#pragma unroll
    for (int i = 0; i < kMB; ++i)
        if (s0 + i < ns) {
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i][tt], acc[tt], 0, 0, 0);
        }
}
// acc[tile] += A[16 envs][K] * Wt[K][16 units of the tile], k ascending (the order of the fmaf chain).  The operands of
// kMB k-steps are fetched together and one block AHEAD of the MFMAs that consume them (two register sets, the loop
// advances by two blocks): with one k-step in flight every MFMA group waited an L2 round trip for its weights
// (~250 ns x 43 steps at H = 128: the first cut of this kernel was no faster than the fmaf chains).
template <int TT>
GX_D void mfma_chain(ps_f4 (&acc)[TT], const float* __restrict__ wt, int H, int col0, const float* A, int AS, int K, int c16, int kq)
{
    const int ns = K >> 2;
    const float* ap = A + c16 * AS + kq;
    const float* bp = wt + (size_t)kq * H + col0 + c16;
    float a0[kMB], b0[kMB][TT], a1[kMB], b1[kMB][TT];
    mfma_fetch<TT>(a0, b0, ap, bp, H, 0, ns);
#pragma unroll 1
    for (int s0 = 0; s0 < ns; s0 += 2 * kMB) {
        mfma_fetch<TT>(a1, b1, ap, bp, H, s0 + kMB, ns);
        mfma_issue<TT>(acc, a0, b0, s0, ns);
        mfma_fetch<TT>(a0, b0, ap, bp, H, s0 + 2 * kMB, ns);
        mfma_issue<TT>(acc, a1, b1, s0 + kMB, ns);
    }
}

template <int H>
__global__ __launch_bounds__(512) void policy_step_mfma_kernel(PolicyStepArgs a)
{
    constexpr int U = H / 64, E = kPmEnv, NT = 512, TT = H / 64, HS = H + 4;
    extern __shared__ float4 ps_lds4[];
    float* lds = reinterpret_cast<float*>(ps_lds4);
    const int D = a.D, A = a.A, Dp = pad4(D), XS = Dp + 1;
    float* X = lds;                              // [E][XS], columns D .. Dp-1 zero
    float* H1 = X + pad4(E * XS);                // [2][E][HS]
    float* H2 = H1 + 2 * E * HS;                 // [2][E][HS]
    float* outs = H2 + 2 * E * HS;               // [E][A + 1]
    const int tid = threadIdx.x, wave = tid >> 6, lw = tid & 63, c16 = lw & 15, kq = lw >> 4;
    const int net = wave >> 2, col0 = 16 * TT * (wave & 3);
    const int env0 = blockIdx.x * E;
    const int msz_pi = H * D + H + H * H + H + A * H + A;
    const float* g = a.params + (net ? msz_pi : 0);
    const float *b1 = g + H * D, *b2 = b1 + H + H * H;
    const float* wt1 = a.wt + (size_t)net * (Dp * H + H * H);
    const float* wt2 = wt1 + Dp * H;

    for (int i = tid; i < E * XS; i += NT) {
        const int e = i / XS, k = i - e * XS;
        const int env = env0 + e;
        float x = 0.0f;
        if (k < D && env < a.N) {
            x = a.obs[(size_t)env * D + k];
            if (a.mode == 0) a.obs_in[(size_t)env * D + k] = x;
            else a.obs_last[(size_t)env * D + k] = x;
        }
        X[i] = x;
    }
    __syncthreads();
    const bool skip = a.mode == 1 && net == 0; // the bootstrap value needs the critic only
    ps_f4 acc[TT];
    if (!skip) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) { const float bb = b1[col0 + 16 * tt + c16]; acc[tt] = ps_f4{bb, bb, bb, bb}; }
        mfma_chain<TT>(acc, wt1, H, col0, X, XS, Dp, c16, kq);
        float* o = H1 + (size_t)net * E * HS + col0 + c16;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(4 * kq + r) * HS + 16 * tt] = tanh_f(acc[tt][r]);
    }
    __syncthreads();
    if (!skip) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) { const float bb = b2[col0 + 16 * tt + c16]; acc[tt] = ps_f4{bb, bb, bb, bb}; }
        mfma_chain<TT>(acc, wt2, H, col0, H1 + (size_t)net * E * HS, HS, H, c16, kq);
        float* o = H2 + (size_t)net * E * HS + col0 + c16;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(4 * kq + r) * HS + 16 * tt] = tanh_f(acc[tt][r]);
    }
    __syncthreads();
    // output layers: task (env e, output o) on 16 lanes; o < A: mu_o (actor), o == A: the value (critic)
    const int l = tid & 15;
    for (int task = tid >> 4; task < E * (A + 1); task += NT / 16) {
        const int e = task / (A + 1), o = task - e * (A + 1);
        const int nt = o == A ? 1 : 0, oo = nt ? 0 : o;
        if (a.mode == 1 && !nt) continue;
        const float* gg = a.params + (nt ? msz_pi : 0);
        const float* W3 = gg + H * D + H + H * H + H;
        const float* b3 = W3 + (nt ? 1 : A) * H;
        const float* hh = H2 + ((size_t)nt * E + e) * HS;
        float pp = 0.0f;
#pragma unroll
        for (int c = 0; c < U; ++c) {
            const float4 hv = *reinterpret_cast<const float4*>(hh + 64 * c + 4 * l);
            const float4 wv = *reinterpret_cast<const float4*>(W3 + oo * H + 64 * c + 4 * l);
            pp = fmaf(hv.x, wv.x, pp); pp = fmaf(hv.y, wv.y, pp); pp = fmaf(hv.z, wv.z, pp); pp = fmaf(hv.w, wv.w, pp);
        }
        pp = pp + __shfl_xor(pp, 8, 16);
        pp = pp + __shfl_xor(pp, 4, 16);
        pp = pp + __shfl_xor(pp, 2, 16);
        pp = pp + __shfl_xor(pp, 1, 16);
        if (l == 0) outs[e * (A + 1) + o] = b3[oo] + pp;
    }
    __syncthreads();
    policy_step_tail<E>(a, outs, env0, tid, H, msz_pi);
}

int policy_step_wt_floats(int D, int H) { return 2 * (pad4(D) * H + H * H); }

void launch_policy_transpose(const float* params, float* wt, int D, int A, int H, hipStream_t s)
{
    const int n = policy_step_wt_floats(D, H);
    hipLaunchKernelGGL(policy_transpose_kernel, dim3((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), dim3(256), 0, s, params, wt,
                       D, A, H);
}

bool policy_step_supported(int H) { return H == 64 || H == 128 || H == 192 || H == 256; }

void launch_policy_step(int H, const float* params, const float* wt, const float* obs, uint32_t seed0, uint32_t seed1,
                        uint32_t tnoise, int N, int D, int A, int env_offset, int mode, float* obs_in, float* act, float* mu,
                        float* logp, float* val, float* obs_last, float* logstd, hipStream_t s, bool valu)
{
    PolicyStepArgs a;
    a.params = params; a.wt = wt; a.obs = obs; a.seed0 = seed0; a.seed1 = seed1; a.tnoise = tnoise;
    a.N = N; a.D = D; a.A = A; a.env_offset = env_offset; a.mode = mode;
    a.obs_in = obs_in; a.act = act; a.mu = mu; a.logp = logp; a.val = val; a.obs_last = obs_last; a.logstd = logstd;
    if (!valu) {
        const dim3 gm((N + kPmEnv - 1) / kPmEnv);
        const size_t lm = sizeof(float) * ((size_t)pad4(kPmEnv * (pad4(D) + 1)) + 4 * (size_t)kPmEnv * (H + 4) +
                                           (size_t)kPmEnv * (A + 1) + 4);
        if (H == 64) hipLaunchKernelGGL((policy_step_mfma_kernel<64>), gm, dim3(512), lm, s, a);
        else if (H == 128) hipLaunchKernelGGL((policy_step_mfma_kernel<128>), gm, dim3(512), lm, s, a);
        else if (H == 192) hipLaunchKernelGGL((policy_step_mfma_kernel<192>), gm, dim3(512), lm, s, a);
        else hipLaunchKernelGGL((policy_step_mfma_kernel<256>), gm, dim3(512), lm, s, a);
        return;
    }
    const dim3 grid((N + kPsEnv - 1) / kPsEnv);
    const size_t lds = sizeof(float) * ((size_t)pad4(D) * kPsEnv + 4 * (size_t)H * kPsEnv + (size_t)kPsEnv * (A + 1) + 4);
    if (H == 64) hipLaunchKernelGGL((policy_step_kernel<64>), grid, dim3(128), lds, s, a);
    else if (H == 128) hipLaunchKernelGGL((policy_step_kernel<128>), grid, dim3(256), lds, s, a);
    else if (H == 192) hipLaunchKernelGGL((policy_step_kernel<192>), grid, dim3(384), lds, s, a);
    else hipLaunchKernelGGL((policy_step_kernel<256>), grid, dim3(512), lds, s, a);
}

} // namespace gx
