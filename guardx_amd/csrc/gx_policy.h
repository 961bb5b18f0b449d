// gx_policy.h -- on-device `ac.step(o)` for the fused closed-loop rollout:
// MLPActorCritic(hidden_sizes=(64,64), tanh) of safe_rl_libX/trpo/trpo_core.py:110-173 --
// Gaussian actor (mu_net + state-independent log_std) and MLP critic -- evaluated by the 16
// lanes that own an environment in the lane-group kernel.
//
// Weights live in LDS, hidden layers transposed to [in][out] so that lane l reads the four
// output columns 4l..4l+3 of one input row as one conflict-free ds_read_b128; the input vector
// (the env's observation row, then the first hidden layer) is read as LDS broadcasts.  Every
// accumulation is a sequential fmaf chain over the inputs; the output layer is 16 lane partials
// folded by a butterfly -- the order oracle/gx_oracle.c:mlp_forward restates.
#pragma once
#include "gx_device.h"

namespace gx {

constexpr int kPolHd = 64; // hidden width (the reference default --hid 64 --l 2)

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() is a release / acquire fence over ALL address
// spaces: the compiler puts s_waitcnt vmcnt(0) in front of the barrier, i.e. every wave first waits until the global
// stores of its step outputs (observation row, action, mu, logp, value, reward, cost, done) have been acknowledged -- an L2
// round trip per barrier, several barriers per control step.  What the waves of a policy workgroup hand each other
// (observation rows, hidden activations) lives in LDS; nothing written to global memory is read again in the kernel.
GX_D void wg_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct PolicyArgs {
    const float* params;   // pi{W1[Hd][D] b1 W2[Hd][Hd] b2 W3[A][Hd] b3} v{.. W3[1][Hd] b3} log_std[A]
    const float* wt;       // widths 192 / 256 (streaming form): the [k][unit] transposed hidden layers (policy_transpose_kernel)
    uint32_t seed0, seed1; // key of the action-noise stream
    uint32_t t0;           // policy steps taken before this launch (noise counter offset)
    const float* obs0;     // [N][D] observation at entry
    float* obs_in;         // [T][N][D] observation the policy saw at step t
    float* act;            // [T][N][A]
    float* logp;           // [T][N]
    float* val;            // [T][N]
    float* mu;             // [T][N][A]
    float* obs_last;       // [N][D] observation after the last step (post reset_done)
    float* val_last;       // [N]    V(obs_last) for the bootstrap
    float* logstd;         // [A]    log(std) as ac.step returns it
};

GX_HD int mlp_lds_floats(int rows, int Out) { return kPolHd * rows + kPolHd + kPolHd * kPolHd + kPolHd + Out * kPolHd + Out; }
GX_HD int mlp_floats(int D, int Out) { return kPolHd * D + kPolHd + kPolHd * kPolHd + kPolHd + Out * kPolHd + Out; }

// LDS image of one network: Wt1[D][Hd] b1[Hd] Wt2[Hd][Hd] b2[Hd] W3[Out][Hd] b3[Out]
struct MlpLds { const float *Wt1, *b1, *Wt2, *b2, *W3, *b3; };

// `rows` = input rows allocated for Wt1 (D, or D padded to a multiple of 4 with zero rows for MFMA)
GX_D MlpLds mlp_lds_view(const float* base, int rows, int Out)
{
    MlpLds m;
    m.Wt1 = base; m.b1 = m.Wt1 + kPolHd * rows; m.Wt2 = m.b1 + kPolHd; m.b2 = m.Wt2 + kPolHd * kPolHd;
    m.W3 = m.b2 + kPolHd; m.b3 = m.W3 + Out * kPolHd;
    return m;
}

// cooperative load of one network from global (torch layout [out][in]) into its LDS image
GX_D void mlp_stage(float* lds, const float* __restrict__ g, int D, int rows, int Out, int tid, int nthreads)
{
    const float* gW1 = g; const float* gb1 = gW1 + kPolHd * D; const float* gW2 = gb1 + kPolHd;
    const float* gb2 = gW2 + kPolHd * kPolHd; const float* gW3 = gb2 + kPolHd; const float* gb3 = gW3 + Out * kPolHd;
    float* Wt1 = lds; float* b1 = Wt1 + kPolHd * rows; float* Wt2 = b1 + kPolHd; float* b2 = Wt2 + kPolHd * kPolHd;
    float* W3 = b2 + kPolHd; float* b3 = W3 + Out * kPolHd;
    for (int i = tid; i < kPolHd * D; i += nthreads) { const int j = i / D, k = i - j * D; Wt1[k * kPolHd + j] = gW1[i]; }
    for (int i = tid + kPolHd * D; i < kPolHd * rows; i += nthreads) Wt1[i] = 0.0f; // zero padding rows
    for (int i = tid; i < kPolHd * kPolHd; i += nthreads) { const int j = i >> 6, k = i & 63; Wt2[k * kPolHd + j] = gW2[i]; }
    for (int i = tid; i < kPolHd; i += nthreads) { b1[i] = gb1[i]; b2[i] = gb2[i]; }
    for (int i = tid; i < Out * kPolHd; i += nthreads) W3[i] = gW3[i];
    for (int i = tid; i < Out; i += nthreads) b3[i] = gb3[i];
}

// one fmaf step of four output columns
#define GX_FMA4(acc, xv, wv)                                                           \
    do { acc.x = fmaf(xv, wv.x, acc.x); acc.y = fmaf(xv, wv.y, acc.y);                 \
         acc.z = fmaf(xv, wv.z, acc.z); acc.w = fmaf(xv, wv.w, acc.w); } while (0)

GX_D float4 tanh4(float4 a) { return make_float4(tanh_f(a.x), tanh_f(a.y), tanh_f(a.z), tanh_f(a.w)); }

// output layer: 16 lane partials over the lane's four hidden units, folded by a butterfly
GX_D float head_out(const MlpLds& w, int o, int l, float4 h)
{
    const float4 wv = *reinterpret_cast<const float4*>(w.W3 + o * kPolHd + 4 * l);
    float pp = 0.0f;
    pp = fmaf(h.x, wv.x, pp); pp = fmaf(h.y, wv.y, pp); pp = fmaf(h.z, wv.z, pp); pp = fmaf(h.w, wv.w, pp);
    pp = pp + __shfl_xor(pp, 8, 16);
    pp = pp + __shfl_xor(pp, 4, 16);
    pp = pp + __shfl_xor(pp, 2, 16);
    pp = pp + __shfl_xor(pp, 1, 16);
    return w.b3[o] + pp;
}

// actor and critic in one pass (both read the same observation): per input k one broadcast read of
// x serves eight fmaf chains; x is fetched four at a time.  x = LDS row of D inputs (16-byte
// aligned), hbuf = LDS [2][Hd] scratch of this env group.  Whole-wave call.  Each output is its own
// sequential fmaf chain over k, so the values equal the one-network-at-a-time evaluation bit for bit.
template <int A>
GX_D void actor_critic_forward(const MlpLds& wp, const MlpLds& wc, const float* x, float* hbuf, int D, int l,
                               float (&mu)[A], float& v)
{
    float4 ap = *reinterpret_cast<const float4*>(wp.b1 + 4 * l);
    float4 ac = *reinterpret_cast<const float4*>(wc.b1 + 4 * l);
    const int D4 = D & ~3;
    for (int k = 0; k < D4; k += 4) {
        const float4 xv = *reinterpret_cast<const float4*>(x + k);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 w1 = *reinterpret_cast<const float4*>(wp.Wt1 + (k + u) * kPolHd + 4 * l);
            const float4 w2 = *reinterpret_cast<const float4*>(wc.Wt1 + (k + u) * kPolHd + 4 * l);
            GX_FMA4(ap, xs[u], w1);
            GX_FMA4(ac, xs[u], w2);
        }
    }
    for (int k = D4; k < D; ++k) {
        const float xv = x[k];
        const float4 w1 = *reinterpret_cast<const float4*>(wp.Wt1 + k * kPolHd + 4 * l);
        const float4 w2 = *reinterpret_cast<const float4*>(wc.Wt1 + k * kPolHd + 4 * l);
        GX_FMA4(ap, xv, w1);
        GX_FMA4(ac, xv, w2);
    }
    *reinterpret_cast<float4*>(hbuf + 4 * l) = tanh4(ap);
    *reinterpret_cast<float4*>(hbuf + kPolHd + 4 * l) = tanh4(ac);
    __syncthreads();
    ap = *reinterpret_cast<const float4*>(wp.b2 + 4 * l);
    ac = *reinterpret_cast<const float4*>(wc.b2 + 4 * l);
#pragma unroll 4
    for (int k = 0; k < kPolHd; k += 4) {
        const float4 xp = *reinterpret_cast<const float4*>(hbuf + k);
        const float4 xc = *reinterpret_cast<const float4*>(hbuf + kPolHd + k);
        const float xps[4] = {xp.x, xp.y, xp.z, xp.w}, xcs[4] = {xc.x, xc.y, xc.z, xc.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 w1 = *reinterpret_cast<const float4*>(wp.Wt2 + (k + u) * kPolHd + 4 * l);
            const float4 w2 = *reinterpret_cast<const float4*>(wc.Wt2 + (k + u) * kPolHd + 4 * l);
            GX_FMA4(ap, xps[u], w1);
            GX_FMA4(ac, xcs[u], w2);
        }
    }
    const float4 hp = tanh4(ap), hc = tanh4(ac);
#pragma unroll
    for (int o = 0; o < A; ++o) mu[o] = head_out(wp, o, l, hp);
    v = head_out(wc, 0, l, hc);
    __syncthreads(); // hbuf is reused by the next call
}

// critic only (bootstrap value of the final observation)
GX_D float critic_forward(const MlpLds& wc, const float* x, float* hbuf, int D, int l)
{
    float4 ac = *reinterpret_cast<const float4*>(wc.b1 + 4 * l);
    for (int k = 0; k < D; ++k) {
        const float xv = x[k];
        const float4 w2 = *reinterpret_cast<const float4*>(wc.Wt1 + k * kPolHd + 4 * l);
        GX_FMA4(ac, xv, w2);
    }
    *reinterpret_cast<float4*>(hbuf + 4 * l) = tanh4(ac);
    __syncthreads();
    ac = *reinterpret_cast<const float4*>(wc.b2 + 4 * l);
    for (int k = 0; k < kPolHd; ++k) {
        const float xv = hbuf[k];
        const float4 w2 = *reinterpret_cast<const float4*>(wc.Wt2 + k * kPolHd + 4 * l);
        GX_FMA4(ac, xv, w2);
    }
    const float v = head_out(wc, 0, l, tanh4(ac));
    __syncthreads();
    return v;
}

// ---------------------------------------------------------------------------
// MFMA form (256-thread workgroup = 4 waves = 16 environments).  Each hidden layer is
// H[16 envs][64 units] = X[16][K] * Wt[K][64] for both networks: 8 output tiles of 16x16, two per
// wave, each a chain of v_mfma_f32_16x16x4_f32 over K.  That instruction accumulates exactly like a
// sequential fmaf chain over k (tools/probes/mfma_f32_probe.hip: 0 mismatches), so the results are
// bit-identical to the VALU form and to the CPU restatement.  Operand layout: A lane = k*16 + env,
// B lane = k*16 + unit, D lane holds envs 4*(lane/16)..+3 of unit lane%16.
// ---------------------------------------------------------------------------
typedef float mfma_f4 __attribute__((ext_vector_type(4)));
constexpr int kPolHS = 68; // LDS row stride of the hidden activations (16-byte aligned rows)

// One hidden layer, two 16x16 output tiles per wave.  KS > 0: the number of k-steps (K / 4) is a compile-time
// constant -- every A / B operand of both tiles is fetched from LDS up front and the two accumulator chains are
// issued alternately, so that neither the LDS latency (two dependent ds_read per MFMA in the round-1 loop: ~170
// cycles per k-step) nor the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32 sits between two MFMAs.  The
// order of accumulation within a tile (k ascending) is unchanged, hence the same bits.  KS == 0: run-time K.
template <int KS>
GX_D void mfma_layer(const MlpLds& wp, const MlpLds& wc, bool second, const float* in, int in_stride,
                     int in_net_stride, int K, float* out, int wave, int lw)
{
    const int c16 = lw & 15, kq = lw >> 4;
    const float* arow[2];
    const float* bcol[2];
    float* o[2];
    mfma_f4 acc[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * wave + tt, net = t >> 2, ut = t & 3;
        const MlpLds& w = net ? wc : wp;
        const float* Wt = second ? w.Wt2 : w.Wt1;
        const float bias = (second ? w.b2 : w.b1)[16 * ut + c16];
        acc[tt] = mfma_f4{bias, bias, bias, bias};
        arow[tt] = in + net * in_net_stride + c16 * in_stride + kq;
        bcol[tt] = Wt + kq * kPolHd + 16 * ut + c16;
        o[tt] = out + net * 16 * kPolHS + 16 * ut + c16;
    }
    if (KS > 0) {
        float av[2][KS > 0 ? KS : 1], bv[2][KS > 0 ? KS : 1];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) { av[tt][s] = arow[tt][4 * s]; bv[tt][s] = bcol[tt][4 * s * kPolHd]; }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][s], bv[0][s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][s], bv[1][s], acc[1], 0, 0, 0);
        }
    } else {
        for (int k0 = 0; k0 < K; k0 += 4) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[0][k0], bcol[0][k0 * kPolHd], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[1][k0], bcol[1][k0 * kPolHd], acc[1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[tt][(4 * kq + r) * kPolHS] = tanh_f(acc[tt][r]);
    }
}

// both hidden layers for the 16 envs of the workgroup; X = [16][XS] observations (zero padded to Kp),
// H1/H2 = [2 nets][16][kPolHS].  Whole-workgroup call.
GX_D void mfma_hidden(const MlpLds& wp, const MlpLds& wc, const float* X, int XS, int Kp, float* H1, float* H2,
                      int wave, int lw)
{
    // the observation widths of the four robots' default tasks (Point 43, Swimmer 46, Ant 64, Walker 70 -> pad4)
    if (Kp == 44) mfma_layer<11>(wp, wc, false, X, XS, 0, Kp, H1, wave, lw);
    else if (Kp == 48) mfma_layer<12>(wp, wc, false, X, XS, 0, Kp, H1, wave, lw);
    else if (Kp == 64) mfma_layer<16>(wp, wc, false, X, XS, 0, Kp, H1, wave, lw);
    else if (Kp == 72) mfma_layer<18>(wp, wc, false, X, XS, 0, Kp, H1, wave, lw);
    else mfma_layer<0>(wp, wc, false, X, XS, 0, Kp, H1, wave, lw);
    wg_sync_lds();
    mfma_layer<kPolHd / 4>(wp, wc, true, H1, kPolHS, 16 * kPolHS, kPolHd, H2, wave, lw);
    wg_sync_lds();
}

// two standard normals from one Threefry block keyed by `seed`, counter (global env, step*16+pair)
GX_D void normal_pair(uint32_t s0, uint32_t s1, uint32_t env, uint32_t ctr, float& z0, float& z1)
{
    uint32_t b0, b1;
    threefry2x32(s0, s1, env, ctr, b0, b1);
    const float u1 = (float)((b0 >> 8) + 1u) * 5.9604644775390625e-08f;
    const float u2 = (float)(b1 >> 8) * 5.9604644775390625e-08f;
    const float r = sqrtf(-2.0f * log_f(u1));
    float sn, cs;
    sincos_f(6.2831854820251465f * u2, sn, cs);
    z0 = r * cs;
    z1 = r * sn;
}

GX_HD int pad4(int n) { return (n + 3) & ~3; }

// ---------------------------------------------------------------------------
// Width 128 in ONE launch (round 5): hidden_sizes = (128, 128) kept ON CHIP across the whole rollout.
// The two networks are 179 KB -- they do not fit the LDS -- but in the MFMA form a lane only ever needs ITS B operands:
// wave w of the 4-wave workgroup owns network w / 2 and the unit tiles 4 (w % 2) .. + 3 (16 units each), i.e. per
// k-step of four inputs ONE float per tile and lane: KS1 x 4 floats for the first layer (KS1 = pad4(D) / 4 k-steps: 11
// for the Point's 43 observations) and 32 x 4 for the second -- 172 registers, loaded once before the step loop from the
// torch-layout parameters.  A workgroup is alone on its CU (125 workgroups of 16 envs on 256 CUs at env_num = 2000), so
// a wave has the whole 512-register file of its SIMD.  LDS holds what the 16-lane groups read: biases, output layers,
// the observation rows and the hidden activations.  Per control step nothing but the env's own outputs moves.
// The arithmetic is the step-wise kernel's (gx_policy_step.hip) and the checker's: every hidden unit one
// v_mfma_f32_16x16x4_f32 chain over k ascending (= a sequential fmaf chain), the output layer 16 lane partials over the
// units 64 c + 4 l + j folded by the same butterfly.
// ---------------------------------------------------------------------------
constexpr int kPolHd2 = 128;          // the width this form serves
constexpr int kPolHS2 = kPolHd2 + 4;  // LDS row stride of its hidden activations
constexpr int kPol2KS = kPolHd2 / 4;  // k-steps of the second layer

// LDS image of one network's small parts: b1[128] b2[128] W3[Out][128] b3[Out]
// (H = the hidden width: kPolHd2 for the register-resident form, 192 / 256 for the streaming form below)
GX_HD int mlp2_head_floats(int Out, int H = kPolHd2) { return 2 * H + Out * H + Out; }
GX_HD int mlp2_floats(int D, int Out, int H = kPolHd2) { return H * D + H + H * H + H + Out * H + Out; }
struct Mlp2Head { const float *b1, *b2, *W3, *b3; };
GX_D Mlp2Head mlp2_head_view(const float* base, int Out, int H = kPolHd2)
{
    Mlp2Head m;
    m.b1 = base; m.b2 = m.b1 + H; m.W3 = m.b2 + H; m.b3 = m.W3 + Out * H;
    return m;
}
GX_D void mlp2_head_stage(float* lds, const float* __restrict__ g, int D, int Out, int tid, int nthreads, int H = kPolHd2)
{
    const float* gb1 = g + H * D; const float* gb2 = gb1 + H + H * H;
    const float* gW3 = gb2 + H; const float* gb3 = gW3 + Out * H;
    float* b1 = lds; float* b2 = b1 + H; float* W3 = b2 + H; float* b3 = W3 + Out * H;
    for (int i = tid; i < H; i += nthreads) { b1[i] = gb1[i]; b2[i] = gb2[i]; }
    for (int i = tid; i < Out * H; i += nthreads) W3[i] = gW3[i];
    for (int i = tid; i < Out; i += nthreads) b3[i] = gb3[i];
}

// this lane's B operands and biases: [k-step][tile]
template <int KS1>
struct Pol2Regs { float w1[KS1][4], w2[kPol2KS][4], bias1[4], bias2[4]; };

template <int KS1>
GX_D void pol2_load(Pol2Regs<KS1>& W, const float* __restrict__ params, int D, int A, int wave, int lw)
{
    const int c16 = lw & 15, kq = lw >> 4, net = wave >> 1, ut0 = 4 * (wave & 1);
    const float* g = params + (net ? mlp2_floats(D, A) : 0);
    const float* W1 = g; const float* b1 = W1 + kPolHd2 * D; const float* W2 = b1 + kPolHd2; const float* b2 = W2 + kPolHd2 * kPolHd2;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        const int unit = 16 * (ut0 + tt) + c16;
        W.bias1[tt] = b1[unit]; W.bias2[tt] = b2[unit];
#pragma unroll
        for (int s = 0; s < KS1; ++s) { const int k = 4 * s + kq; W.w1[s][tt] = k < D ? W1[(size_t)unit * D + k] : 0.0f; }
#pragma unroll
        for (int s = 0; s < kPol2KS; ++s) W.w2[s][tt] = W2[(size_t)unit * kPolHd2 + 4 * s + kq];
    }
}

// both hidden layers for the 16 envs of the workgroup: X = [16][XS] observations (columns D .. 4 KS1 - 1 zero),
// H1 / H2 = [2 nets][16][kPolHS2].  Whole-workgroup call (two barriers).
template <int KS1>
GX_D void pol2_hidden(const Pol2Regs<KS1>& W, const float* X, int XS, float* H1, float* H2, int wave, int lw)
{
    const int c16 = lw & 15, kq = lw >> 4, net = wave >> 1, ut0 = 4 * (wave & 1);
    mfma_f4 acc[4];
    {
        float av[KS1];
        const float* ap = X + c16 * XS + kq;
#pragma unroll
        for (int s = 0; s < KS1; ++s) av[s] = ap[4 * s];
        __builtin_amdgcn_sched_barrier(0); // (the scheduler otherwise sinks each ds_read to its MFMAs: an LDS round trip per 8)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt] = mfma_f4{W.bias1[tt], W.bias1[tt], W.bias1[tt], W.bias1[tt]};
#pragma unroll
        for (int s = 0; s < KS1; ++s)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], W.w1[s][tt], acc[tt], 0, 0, 0);
        float* o = H1 + (size_t)net * 16 * kPolHS2 + 16 * ut0 + c16;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(4 * kq + r) * kPolHS2 + 16 * tt] = tanh_f(acc[tt][r]);
    }
    wg_sync_lds();
    {
        float av[kPol2KS];
        const float* ap = H1 + (size_t)net * 16 * kPolHS2 + c16 * kPolHS2 + kq;
#pragma unroll
        for (int s = 0; s < kPol2KS; ++s) av[s] = ap[4 * s];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt] = mfma_f4{W.bias2[tt], W.bias2[tt], W.bias2[tt], W.bias2[tt]};
#pragma unroll
        for (int s = 0; s < kPol2KS; ++s)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], W.w2[s][tt], acc[tt], 0, 0, 0);
        float* o = H2 + (size_t)net * 16 * kPolHS2 + 16 * ut0 + c16;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(4 * kq + r) * kPolHS2 + 16 * tt] = tanh_f(acc[tt][r]);
    }
    wg_sync_lds();
}

// output layer of the wide networks: partial l over the units 64 c + 4 l + j (c = 0 .. H / 64 - 1), butterfly, + bias
// (gx_policy_step.hip, oracle/gx_oracle.c:mlp_forward)
template <int H = kPolHd2>
GX_D float head2_out(const Mlp2Head& w, int o, int l, const float* h2row)
{
    float pp = 0.0f;
#pragma unroll
    for (int c = 0; c < H / 64; ++c) {
        const float4 hv = *reinterpret_cast<const float4*>(h2row + 64 * c + 4 * l);
        const float4 wv = *reinterpret_cast<const float4*>(w.W3 + o * H + 64 * c + 4 * l);
        pp = fmaf(hv.x, wv.x, pp); pp = fmaf(hv.y, wv.y, pp); pp = fmaf(hv.z, wv.z, pp); pp = fmaf(hv.w, wv.w, pp);
    }
    pp = pp + __shfl_xor(pp, 8, 16);
    pp = pp + __shfl_xor(pp, 4, 16);
    pp = pp + __shfl_xor(pp, 2, 16);
    pp = pp + __shfl_xor(pp, 1, 16);
    return w.b3[o] + pp;
}

// dynamic LDS of the policy variants, in floats.
//  VALU form (64 threads, 4 envs):   pi image | v image | log_std,std | hbuf[4][2][Hd] | xrow[4][pad4 D]
//  MFMA form (256 threads, 16 envs): pi image | v image (Wt1 zero-padded to pad4 D rows) | log_std,std |
//                                    X[16][pad4 D + 1] | H1[2][16][68] | H2[2][16][68]
// ---------------------------------------------------------------------------
// Widths 192 and 256 in ONE launch (round 5): the hidden-layer weights (0.35 / 0.61 MB for both networks) fit neither
// the LDS nor the registers, but every workgroup reads the SAME [k][unit] transposed copy (policy_transpose_kernel,
// gx_policy_step.hip), which stays in the L2 of its XCD: each k-step's B operands are streamed from there, kSB k-steps
// at a time and one block ahead of the MFMAs that consume them, while the A operands (observation rows, first hidden
// layer) come from LDS.  What the step-wise form paid per control step -- two kernel launches, the observation's
// round trip through global memory, a cold start of every wave -- is gone; the arithmetic (k ascending per unit) is the same.
// Wave w of the 4-wave workgroup: network w / 2, unit tiles (H / 32) (w % 2) .. + H / 32 - 1.
// ---------------------------------------------------------------------------
constexpr int kSB = 8; // k-steps (of 4 inputs) whose operands are in flight together: 64 MFMAs at H = 256, about one L2 round trip
// Tile tt of a wave holds the units col0 + TT c + tt (c = 0 .. 15): lane (kq, c) needs, per k-step, the TT CONSECUTIVE
// floats Wt[k][col0 + TT c ..] -- two 16-byte loads at H = 256, three 8-byte loads at H = 192, and the 16 lanes of a k row
// read one contiguous 512 / 384 bytes (tiles of 16 adjacent units would take TT 4-byte loads per k-step, 64 bytes per row
// each).  Which unit sits in which tile slot changes nothing: every unit is its own accumulation chain.
template <int TT>
GX_D void polS_fetch(float (&av)[kSB], float (&bv)[kSB][TT], const float* ap, const float* bp, int H, int s0, int ns)
{
#pragma unroll
    for (int i = 0; i < kSB; ++i) {
        const int sidx = s0 + i;
        if (sidx < ns) { // wave-uniform
            av[i] = ap[4 * sidx];
            const float* row = bp + (size_t)(4 * sidx) * H;
            if constexpr (TT % 4 == 0) {
#pragma unroll
                for (int q = 0; q < TT / 4; ++q) {
                    const float4 w4 = *reinterpret_cast<const float4*>(row + 4 * q);
                    bv[i][4 * q] = w4.x; bv[i][4 * q + 1] = w4.y; bv[i][4 * q + 2] = w4.z; bv[i][4 * q + 3] = w4.w;
                }
            } else {
                static_assert(TT % 2 == 0, "even number of tiles per wave");
#pragma unroll
                for (int q = 0; q < TT / 2; ++q) {
                    const float2 w2 = *reinterpret_cast<const float2*>(row + 2 * q);
                    bv[i][2 * q] = w2.x; bv[i][2 * q + 1] = w2.y;
                }
            }
        }
    }
}
template <int TT>
GX_D void polS_issue(mfma_f4 (&acc)[TT], const float (&av)[kSB], const float (&bv)[kSB][TT], int s0, int ns)
{
#pragma unroll
    for (int i = 0; i < kSB; ++i)
        if (s0 + i < ns) {
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i][tt], acc[tt], 0, 0, 0);
        }
}
// acc[tile] += A[16 envs][K] Wt[K][units of the tile], k ascending
template <int TT>
GX_D void polS_chain(mfma_f4 (&acc)[TT], const float* __restrict__ wt, int H, int col0, const float* A, int AS, int K, int c16, int kq)
{
    const int ns = K >> 2;
    const float* ap = A + c16 * AS + kq;
    const float* bp = wt + (size_t)kq * H + col0 + TT * c16;
    float a0[kSB], b0[kSB][TT], a1[kSB], b1[kSB][TT];
    polS_fetch<TT>(a0, b0, ap, bp, H, 0, ns);
    // (scheduling barriers: left alone, the machine scheduler sinks every load down to the MFMA that consumes it -- the
    // kernel is short of registers -- and each group of MFMAs then waits an L2 round trip: s_waitcnt vmcnt(0) in front of it)
#pragma unroll 1
    for (int s0 = 0; s0 < ns; s0 += 2 * kSB) {
        __builtin_amdgcn_sched_barrier(0);
        polS_fetch<TT>(a1, b1, ap, bp, H, s0 + kSB, ns);
        __builtin_amdgcn_sched_barrier(0);
        polS_issue<TT>(acc, a0, b0, s0, ns);
        __builtin_amdgcn_sched_barrier(0);
        polS_fetch<TT>(a0, b0, ap, bp, H, s0 + 2 * kSB, ns);
        __builtin_amdgcn_sched_barrier(0);
        polS_issue<TT>(acc, a1, b1, s0 + kSB, ns);
    }
}
// tanh of a wave's accumulators into the activation rows: lane (kq, c) holds envs 4 kq .. + 3 of the units col0 + TT c + tt
template <int TT>
GX_D void polS_store(const mfma_f4 (&acc)[TT], float* o, int HS, int c16, int kq)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float* row = o + (4 * kq + r) * HS + TT * c16;
        if constexpr (TT % 4 == 0) {
#pragma unroll
            for (int q = 0; q < TT / 4; ++q)
                *reinterpret_cast<float4*>(row + 4 * q) = make_float4(tanh_f(acc[4 * q][r]), tanh_f(acc[4 * q + 1][r]),
                                                                      tanh_f(acc[4 * q + 2][r]), tanh_f(acc[4 * q + 3][r]));
        } else {
#pragma unroll
            for (int q = 0; q < TT / 2; ++q)
                *reinterpret_cast<float2*>(row + 2 * q) = make_float2(tanh_f(acc[2 * q][r]), tanh_f(acc[2 * q + 1][r]));
        }
    }
}
// both hidden layers for the 16 envs of the workgroup.  wt = [pi Wt1 | pi Wt2 | v Wt1 | v Wt2] (Wt1 rows padded to Dp);
// hp / hc = the LDS head images (biases).  X = [16][XS], H1 / H2 = [2][16][H + 4].  Whole-workgroup call.
template <int H>
GX_D void polS_hidden(const float* __restrict__ wt, const Mlp2Head& hp, const Mlp2Head& hc, const float* X, int XS, int Dp,
                      float* H1, float* H2, int wave, int lw)
{
    constexpr int TT = H / 32, HS = H + 4;
    const int c16 = lw & 15, kq = lw >> 4, net = wave >> 1, col0 = 16 * TT * (wave & 1);
    const float* wt1 = wt + (size_t)net * (Dp * H + H * H);
    const float* wt2 = wt1 + (size_t)Dp * H;
    const Mlp2Head& hd = net ? hc : hp;
    mfma_f4 acc[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) { const float bb = hd.b1[col0 + TT * c16 + tt]; acc[tt] = mfma_f4{bb, bb, bb, bb}; }
    polS_chain<TT>(acc, wt1, H, col0, X, XS, Dp, c16, kq);
    polS_store<TT>(acc, H1 + (size_t)net * 16 * HS + col0, HS, c16, kq);
    wg_sync_lds();
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) { const float bb = hd.b2[col0 + TT * c16 + tt]; acc[tt] = mfma_f4{bb, bb, bb, bb}; }
    polS_chain<TT>(acc, wt2, H, col0, H1 + (size_t)net * 16 * HS, HS, H, c16, kq);
    polS_store<TT>(acc, H2 + (size_t)net * 16 * HS + col0, HS, c16, kq);
    wg_sync_lds();
}

GX_HD int policy_lds_floats(int D, int A, int pol)
{
    if (pol == 192 || pol == 256) // streaming form of that width: pi head | v head | log_std,std | X | H1 | H2
        return pad4(mlp2_head_floats(A, pol)) + pad4(mlp2_head_floats(1, pol)) + pad4(2 * A) + 16 * (pad4(D) + 1) + 3 +
               2 * 2 * 16 * (pol + 4);
    if (pol == 3) // width 128, register-resident hidden weights: pi head | v head | log_std,std | X | H1 | H2
        return pad4(mlp2_head_floats(A)) + pad4(mlp2_head_floats(1)) + pad4(2 * A) + 16 * (pad4(D) + 1) + 3 +
               2 * 2 * 16 * kPolHS2;
    if (pol == 2)
        return pad4(mlp_lds_floats(pad4(D), A)) + pad4(mlp_lds_floats(pad4(D), 1)) + pad4(2 * A) +
               16 * (pad4(D) + 1) + 3 + 2 * 2 * 16 * kPolHS;
    return pad4(mlp_floats(D, A)) + pad4(mlp_floats(D, 1)) + pad4(2 * A) + 4 * 2 * kPolHd + 4 * pad4(D);
}

} // namespace gx
