// gx_gae.hip -- device side of the learner's rollout buffer:
// TRPOBufferX.store / finish_path / get of safe_rl_libX/trpo/trpo.py:24-146.
//
// The reference buffer is env-major (env_num, max_ep_len, .) torch tensors; its
// finish_path() pulls rewards/values to the host and runs scipy.lfilter per env in a
// Python loop whenever any env finishes (trpo.py:101-119), and get() normalises the
// advantages per env on the host (trpo.py:131-135).  Here both are one kernel launch
// on the caller's stream, with no host synchronisation.
#include "../../include/guardx.h"
#include <hip/hip_runtime.h>
#include <string>

extern gx_status gx_fail_msg(gx_status st, const char* msg); // gx_api.hip

namespace {

// store(): one step of every field into column `ptr` of the env-major buffers  (trpo.py:49-64)
__global__ void store_kernel(int N, int T, int ptr, int obs_dim, int act_dim,
                             const float* __restrict__ obs, const float* __restrict__ act,
                             const float* __restrict__ rew, const float* __restrict__ val,
                             const float* __restrict__ logp, const float* __restrict__ mu,
                             const float* __restrict__ logstd, float* __restrict__ obs_buf,
                             float* __restrict__ act_buf, float* __restrict__ rew_buf,
                             float* __restrict__ val_buf, float* __restrict__ logp_buf,
                             float* __restrict__ mu_buf, float* __restrict__ logstd_buf)
{
    const int W = obs_dim + 3 * act_dim + 3; // floats written per env
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)N * W) return;
    const int env = (int)(gid / W), k = (int)(gid % W);
    const size_t col = (size_t)env * T + ptr;
    if (k < obs_dim) { obs_buf[col * obs_dim + k] = obs[(size_t)env * obs_dim + k]; return; }
    int j = k - obs_dim;
    if (j < act_dim) { act_buf[col * act_dim + j] = act[(size_t)env * act_dim + j]; return; }
    j -= act_dim;
    if (j < act_dim) { mu_buf[col * act_dim + j] = mu[(size_t)env * act_dim + j]; return; }
    j -= act_dim;
    if (j < act_dim) { logstd_buf[col * act_dim + j] = logstd[(size_t)env * act_dim + j]; return; }
    j -= act_dim;
    if (j == 0) rew_buf[col] = rew[env];
    else if (j == 1) val_buf[col] = val[env];
    else logp_buf[col] = logp[env];
}

// finish_path(): GAE-lambda advantages and rewards-to-go over [path_start, ptr) of every env
// whose `done` is set (all envs when done == nullptr).  deltas in fp32 in the reference's
// operand order, the two discounted cumulative sums in fp64 like scipy.lfilter
// (y[n] = x[n] + discount*y[n-1]), cast to fp32 at the end  (trpo.py:85-119, trpo_core.py:42-58).
__global__ void finish_path_kernel(int N, int T, int ptr, const float* __restrict__ rew,
                                   const float* __restrict__ val, const float* __restrict__ last_val,
                                   const float* __restrict__ done, int* __restrict__ path_start,
                                   float gamma, double dg, double dgl, float* __restrict__ adv,
                                   float* __restrict__ ret, int advance)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    if (done && !(done[env] == 1.0f)) return; // np.where(done == 1)  trpo.py:104
    const int s0 = path_start[env];
    const float lv = last_val[env];
    const size_t base = (size_t)env * T;
    double a = 0.0, r = (double)lv; // lfilter state; rews has last_val appended
    float vnext = lv;
    for (int t = ptr - 1; t >= s0; --t) {
        const float rt = rew[base + t], vt = val[base + t];
        const float delta = (rt + gamma * vnext) - vt; // fp32: rews[:-1] + gamma*vals[1:] - vals[:-1]
        a = (double)delta + dgl * a;
        r = (double)rt + dg * r;
        adv[base + t] = (float)a;
        ret[base + t] = (float)r;
        vnext = vt;
    }
    if (advance) path_start[env] = ptr; // trpo.py:119
}

// get(): per-env advantage normalisation (x - mean) / std, population std, no epsilon
// (trpo.py:131-135 via mpi_statistics_scalar).  One wave per env.
__global__ void adv_normalize_kernel(int N, int T, float* __restrict__ adv, int scale)
{
    const int env = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (env >= N) return;
    float* row = adv + (size_t)env * T;
    float s = 0.0f;
    for (int t = lane; t < T; t += 64) s += row[t];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)T;
    float q = 0.0f;
    for (int t = lane; t < T; t += 64) { const float d = row[t] - mean; q += d * d; }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float sd = sqrtf(q / (float)T);
    if (scale) { for (int t = lane; t < T; t += 64) row[t] = (row[t] - mean) / sd; }
    else { for (int t = lane; t < T; t += 64) row[t] = row[t] - mean; } // cpo.py:158-162: centred, not scaled
}

// GAE over a whole fused rollout (time-major [T][N] arrays): the same arithmetic as finish_path,
// with a path ending wherever done[t] == 1 (bootstrap 0, trpo.py:530-531) and at the end of the tape
// (bootstrap `last_val[env]`, which the caller zeroes when it mirrors trpo.py:506-515).
__global__ void gae_rollout_kernel(int N, int T, const float* __restrict__ rew, const float* __restrict__ val,
                                   const float* __restrict__ done, const float* __restrict__ last_val,
                                   float gamma, double dg, double dgl, float* __restrict__ adv,
                                   float* __restrict__ ret)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    double a = 0.0, r = (double)last_val[env];
    float vnext = last_val[env];
    for (int t = T - 1; t >= 0; --t) {
        const size_t i = (size_t)t * N + env;
        if (done[i] == 1.0f) { a = 0.0; r = 0.0; vnext = 0.0f; } // step t closes a path
        const float rt = rew[i], vt = val[i];
        const float delta = (rt + gamma * vnext) - vt;
        a = (double)delta + dgl * a;
        r = (double)rt + dg * r;
        adv[i] = (float)a;
        ret[i] = (float)r;
        vnext = vt;
    }
}

} // namespace

extern "C" gx_status gx_buffer_store(int32_t env_num, int32_t max_ep_len, int32_t ptr, int32_t obs_dim,
                                     int32_t act_dim, const float* d_obs, const float* d_act,
                                     const float* d_rew, const float* d_val, const float* d_logp,
                                     const float* d_mu, const float* d_logstd, float* d_obs_buf,
                                     float* d_act_buf, float* d_rew_buf, float* d_val_buf,
                                     float* d_logp_buf, float* d_mu_buf, float* d_logstd_buf, void* stream)
{
    if (env_num < 1 || max_ep_len < 1 || ptr < 0 || ptr >= max_ep_len || obs_dim < 1 || act_dim < 1)
        return gx_fail_msg(GX_ERR_ARG, "gx_buffer_store: bad sizes (ptr must be < max_ep_len, trpo.py:56)");
    if (!d_obs || !d_act || !d_rew || !d_val || !d_logp || !d_mu || !d_logstd || !d_obs_buf || !d_act_buf ||
        !d_rew_buf || !d_val_buf || !d_logp_buf || !d_mu_buf || !d_logstd_buf)
        return gx_fail_msg(GX_ERR_ARG, "gx_buffer_store: null pointer");
    const long long total = (long long)env_num * (obs_dim + 3 * act_dim + 3);
    hipLaunchKernelGGL(store_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       env_num, max_ep_len, ptr, obs_dim, act_dim, d_obs, d_act, d_rew, d_val, d_logp, d_mu,
                       d_logstd, d_obs_buf, d_act_buf, d_rew_buf, d_val_buf, d_logp_buf, d_mu_buf, d_logstd_buf);
    return hipGetLastError() == hipSuccess ? GX_OK : gx_fail_msg(GX_ERR_HIP, "gx_buffer_store launch failed");
}

extern "C" gx_status gx_gae_finish_path(int32_t env_num, int32_t max_ep_len, int32_t ptr, const float* d_rew_buf,
                                        const float* d_val_buf, const float* d_last_val, const float* d_done,
                                        int32_t* d_path_start, double gamma, double lam, float* d_adv_buf,
                                        float* d_ret_buf, int32_t advance_path_start, void* stream)
{
    if (env_num < 1 || max_ep_len < 1 || ptr < 0 || ptr > max_ep_len)
        return gx_fail_msg(GX_ERR_ARG, "gx_gae_finish_path: bad sizes");
    if (!d_rew_buf || !d_val_buf || !d_last_val || !d_path_start || !d_adv_buf || !d_ret_buf)
        return gx_fail_msg(GX_ERR_ARG, "gx_gae_finish_path: null pointer");
    // python floats in the reference: the lfilter coefficients gamma and gamma*lam are doubles, the
    // fp32 delta uses gamma rounded to fp32 (weak-scalar promotion)
    hipLaunchKernelGGL(finish_path_kernel, dim3((env_num + 63) / 64), dim3(64), 0, (hipStream_t)stream, env_num,
                       max_ep_len, ptr, d_rew_buf, d_val_buf, d_last_val, d_done, d_path_start, (float)gamma, gamma,
                       gamma * lam, d_adv_buf, d_ret_buf, advance_path_start);
    return hipGetLastError() == hipSuccess ? GX_OK : gx_fail_msg(GX_ERR_HIP, "gx_gae_finish_path launch failed");
}

extern "C" gx_status gx_adv_normalize(int32_t env_num, int32_t max_ep_len, float* d_adv_buf, int32_t scale,
                                      void* stream)
{
    if (env_num < 1 || max_ep_len < 1 || !d_adv_buf) return gx_fail_msg(GX_ERR_ARG, "gx_adv_normalize: bad argument");
    hipLaunchKernelGGL(adv_normalize_kernel, dim3((env_num + 3) / 4), dim3(256), 0, (hipStream_t)stream, env_num,
                       max_ep_len, d_adv_buf, scale);
    return hipGetLastError() == hipSuccess ? GX_OK : gx_fail_msg(GX_ERR_HIP, "gx_adv_normalize launch failed");
}

extern "C" gx_status gx_gae_rollout(int32_t env_num, int32_t T, const float* d_rew, const float* d_val,
                                    const float* d_done, const float* d_last_val, double gamma, double lam,
                                    float* d_adv, float* d_ret, void* stream)
{
    if (env_num < 1 || T < 1 || !d_rew || !d_val || !d_done || !d_last_val || !d_adv || !d_ret)
        return gx_fail_msg(GX_ERR_ARG, "gx_gae_rollout: bad argument");
    hipLaunchKernelGGL(gae_rollout_kernel, dim3((env_num + 63) / 64), dim3(64), 0, (hipStream_t)stream, env_num, T,
                       d_rew, d_val, d_done, d_last_val, (float)gamma, gamma, gamma * lam, d_adv, d_ret);
    return hipGetLastError() == hipSuccess ? GX_OK : gx_fail_msg(GX_ERR_HIP, "gx_gae_rollout launch failed");
}
