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

struct PolicyArgs {
    const float* params;   // pi{W1[Hd][D] b1 W2[Hd][Hd] b2 W3[A][Hd] b3} v{.. W3[1][Hd] b3} log_std[A]
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

GX_HD int mlp_floats(int D, int Out) { return kPolHd * D + kPolHd + kPolHd * kPolHd + kPolHd + Out * kPolHd + Out; }

// LDS image of one network: Wt1[D][Hd] b1[Hd] Wt2[Hd][Hd] b2[Hd] W3[Out][Hd] b3[Out]
struct MlpLds { const float *Wt1, *b1, *Wt2, *b2, *W3, *b3; };

GX_D MlpLds mlp_lds_view(const float* base, int D, int Out)
{
    MlpLds m;
    m.Wt1 = base; m.b1 = m.Wt1 + kPolHd * D; m.Wt2 = m.b1 + kPolHd; m.b2 = m.Wt2 + kPolHd * kPolHd;
    m.W3 = m.b2 + kPolHd; m.b3 = m.W3 + Out * kPolHd;
    return m;
}

// cooperative load of one network from global (torch layout [out][in]) into its LDS image
GX_D void mlp_stage(float* lds, const float* __restrict__ g, int D, int Out, int tid, int nthreads)
{
    const float* gW1 = g; const float* gb1 = gW1 + kPolHd * D; const float* gW2 = gb1 + kPolHd;
    const float* gb2 = gW2 + kPolHd * kPolHd; const float* gW3 = gb2 + kPolHd; const float* gb3 = gW3 + Out * kPolHd;
    float* Wt1 = lds; float* b1 = Wt1 + kPolHd * D; float* Wt2 = b1 + kPolHd; float* b2 = Wt2 + kPolHd * kPolHd;
    float* W3 = b2 + kPolHd; float* b3 = W3 + Out * kPolHd;
    for (int i = tid; i < kPolHd * D; i += nthreads) { const int j = i / D, k = i - j * D; Wt1[k * kPolHd + j] = gW1[i]; }
    for (int i = tid; i < kPolHd * kPolHd; i += nthreads) { const int j = i >> 6, k = i & 63; Wt2[k * kPolHd + j] = gW2[i]; }
    for (int i = tid; i < kPolHd; i += nthreads) { b1[i] = gb1[i]; b2[i] = gb2[i]; }
    for (int i = tid; i < Out * kPolHd; i += nthreads) W3[i] = gW3[i];
    for (int i = tid; i < Out; i += nthreads) b3[i] = gb3[i];
}

// forward pass for the env group of this lane; x = LDS row of D inputs, hbuf = LDS [Hd] scratch of
// the group.  Must be called by the whole wave.  out[o] is identical on the 16 lanes.
template <int OUTMAX>
GX_D void mlp_forward(const MlpLds& w, const float* x, float* hbuf, int D, int Out, int l, float (&out)[OUTMAX])
{
    const float4 bb1 = *reinterpret_cast<const float4*>(w.b1 + 4 * l);
    float a0 = bb1.x, a1 = bb1.y, a2 = bb1.z, a3 = bb1.w;
    for (int k = 0; k < D; ++k) {
        const float xv = x[k];
        const float4 wv = *reinterpret_cast<const float4*>(w.Wt1 + k * kPolHd + 4 * l);
        a0 = fmaf(xv, wv.x, a0); a1 = fmaf(xv, wv.y, a1); a2 = fmaf(xv, wv.z, a2); a3 = fmaf(xv, wv.w, a3);
    }
    *reinterpret_cast<float4*>(hbuf + 4 * l) = make_float4(tanh_f(a0), tanh_f(a1), tanh_f(a2), tanh_f(a3));
    __syncthreads();
    const float4 bb2 = *reinterpret_cast<const float4*>(w.b2 + 4 * l);
    a0 = bb2.x; a1 = bb2.y; a2 = bb2.z; a3 = bb2.w;
#pragma unroll 8
    for (int k = 0; k < kPolHd; ++k) {
        const float xv = hbuf[k];
        const float4 wv = *reinterpret_cast<const float4*>(w.Wt2 + k * kPolHd + 4 * l);
        a0 = fmaf(xv, wv.x, a0); a1 = fmaf(xv, wv.y, a1); a2 = fmaf(xv, wv.z, a2); a3 = fmaf(xv, wv.w, a3);
    }
    const float h0 = tanh_f(a0), h1 = tanh_f(a1), h2 = tanh_f(a2), h3 = tanh_f(a3);
#pragma unroll
    for (int o = 0; o < OUTMAX; ++o) {
        float pp = 0.0f;
        if (o < Out) {
            const float4 wv = *reinterpret_cast<const float4*>(w.W3 + o * kPolHd + 4 * l);
            pp = fmaf(h0, wv.x, pp); pp = fmaf(h1, wv.y, pp); pp = fmaf(h2, wv.z, pp); pp = fmaf(h3, wv.w, pp);
        }
        pp = pp + __shfl_xor(pp, 8, 16);
        pp = pp + __shfl_xor(pp, 4, 16);
        pp = pp + __shfl_xor(pp, 2, 16);
        pp = pp + __shfl_xor(pp, 1, 16);
        out[o] = (o < Out) ? w.b3[o] + pp : 0.0f;
    }
    __syncthreads(); // hbuf is reused by the next network
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

} // namespace gx
