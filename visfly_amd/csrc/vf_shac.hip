// SHAC glue kernels (utils/algorithms/shac.py:215-278 on utils/policies/td_policies.py:82-252): everything of one SHAC
// iteration that is neither the env step / its adjoint (vf_env.hip, vf_env_bwd.hip), the MLP forward / backward (vf_ppo.hip)
// nor TD-lambda (k_td_returns in vf_ppo.hip):
//   k_shac_head_fwd / _bwd   the Actor's action head with its STATE-DEPENDENT log_std (td_policies.py:230-243):
//                            a = tanh(mu + exp(clamp(log_std, -10, 2)) * eps) and its reverse
//   k_shac_accumulate        actor-loss bookkeeping of one horizon step incl. the bootstrap term (shac.py:247-257) and the
//                            horizon-buffer rows the critic update needs (:259-266)
//   k_twin_q_loss            mse_loss(returns, min(Q1, Q2)) and its gradient w.r.t. the two heads (:267-270)
//   k_polyak                 target <- (1 - tau) target + tau param (:274)
// One thread per row; HBM-bound elementwise work (a few dozen bytes per row), no LDS.
#include "vf_common.hpp"

namespace vf {

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

__global__ __launch_bounds__(kBlock) void k_shac_head_fwd(const float4* __restrict__ mu, const float4* __restrict__ log_std,
                                                          const float4* __restrict__ eps, float4* __restrict__ action, int N,
                                                          float lo, float hi)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const float4 m = mu[i], s = log_std[i], e = eps[i];
    // Normal(mean, exp(log_std)).rsample() = loc + eps * scale, then tanh (SB3 SquashedDiagGaussianDistribution.sample)
    action[i] = make_float4(tanhf(m.x + e.x * expf(clampf(s.x, lo, hi))), tanhf(m.y + e.y * expf(clampf(s.y, lo, hi))),
                            tanhf(m.z + e.z * expf(clampf(s.z, lo, hi))), tanhf(m.w + e.w * expf(clampf(s.w, lo, hi))));
}

// d_pre = d_a (1 - a^2);  d_mu = d_pre;  d_log_std = d_pre eps exp(log_std) where lo <= log_std <= hi (torch.clamp passes the
// gradient on the closed interval), else 0
__global__ __launch_bounds__(kBlock) void k_shac_head_bwd(const float4* __restrict__ d_action, const float4* __restrict__ action,
                                                          const float4* __restrict__ log_std, const float4* __restrict__ eps,
                                                          float4* __restrict__ d_mu, float4* __restrict__ d_log_std, int N,
                                                          float lo, float hi)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const float4 da = d_action[i], a = action[i], s = log_std[i], e = eps[i];
    const float4 dp = make_float4(da.x * (1.0f - a.x * a.x), da.y * (1.0f - a.y * a.y), da.z * (1.0f - a.z * a.z),
                                  da.w * (1.0f - a.w * a.w));
    d_mu[i] = dp;
    auto dls = [&](float d, float ee, float ss) { return (ss >= lo && ss <= hi) ? d * ee * expf(ss) : 0.0f; };
    d_log_std[i] = make_float4(dls(dp.x, e.x, s.x), dls(dp.y, e.y, s.y), dls(dp.z, e.z, s.z), dls(dp.w, e.w, s.w));
}

// shac.py:247-257 for one horizon step, per agent:
//   next_value = min(Q1', Q2')                                   (target critics on the detached next observation / action)
//   loss      <- loss - reward * disc
//   cut        = (done | last step of the horizon) & ~episode_done
//   loss      <- loss - next_value * disc * gamma * cut
//   d_reward   = -disc * scale                                    (dLoss / d reward_t of the mean over all agents)
//   disc      <- disc * gamma * ~done + done
// and the buffer rows of the step: next_value, episode_done = done & (ep_flags & VF_EP_EPISODE_DONE)
__global__ __launch_bounds__(kBlock) void k_shac_accumulate(const float* __restrict__ reward, const uint8_t* __restrict__ done,
                                                            const uint8_t* __restrict__ ep_flags, const float* __restrict__ q0,
                                                            const float* __restrict__ q1, float* __restrict__ disc,
                                                            float* __restrict__ loss, float* __restrict__ d_reward,
                                                            float* __restrict__ next_value_row, uint8_t* __restrict__ ep_done_row,
                                                            float gamma, float scale, int last_step, int N)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const float d = disc[i], r = reward[i];
    const bool dn = done[i] != 0;
    const bool epd = dn && (ep_flags[i] & VF_EP_EPISODE_DONE) != 0;
    const float nv = fminf(q0[i], q1[i]);
    float l = loss[i] - r * d;
    const bool cut = (dn || last_step) && !epd;
    l = l - nv * d * gamma * (cut ? 1.0f : 0.0f);
    loss[i] = l;
    d_reward[i] = -d * scale;
    disc[i] = d * gamma * (dn ? 0.0f : 1.0f) + (dn ? 1.0f : 0.0f);
    next_value_row[i] = nv;
    ep_done_row[i] = epd ? 1 : 0;
}

// k_shac_accumulate for the H steps of a recorded horizon in one launch: a thread walks its agent's rows t = 0 .. H-1 in order (the
// discount / loss recurrence is per agent), rows (H, N); the same operations on the same values as H launches
__global__ __launch_bounds__(kBlock) void k_shac_accumulate_horizon(const float* __restrict__ reward, const uint8_t* __restrict__ done,
                                                                    const uint8_t* __restrict__ ep_flags, const float* __restrict__ q0,
                                                                    const float* __restrict__ q1, float* __restrict__ disc,
                                                                    float* __restrict__ loss, float* __restrict__ d_reward,
                                                                    float* __restrict__ next_value, uint8_t* __restrict__ ep_done,
                                                                    float gamma, float scale, int H, int N)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    float d = disc[i], l = loss[i];
    for (int t = 0; t < H; ++t) {
        const size_t j = (size_t)t * N + i;
        const float r = reward[j];
        const bool dn = done[j] != 0;
        const bool epd = dn && (ep_flags[j] & VF_EP_EPISODE_DONE) != 0;
        const float nv = fminf(q0[j], q1[j]);
        l = l - r * d;
        const bool cut = (dn || t == H - 1) && !epd;
        l = l - nv * d * gamma * (cut ? 1.0f : 0.0f);
        d_reward[j] = -d * scale;
        d = d * gamma * (dn ? 0.0f : 1.0f) + (dn ? 1.0f : 0.0f);
        next_value[j] = nv;
        ep_done[j] = epd ? 1 : 0;
    }
    disc[i] = d;
    loss[i] = l;
}

// values = min(Q1, Q2) (ties: the first, like torch.min over dim 1); loss = mean((returns - values)^2);
// dQ_k = 2 (values - returns) / M_global where Q_k is the minimum, else 0.  Per-block fp64 partial sums of the squared error.
__global__ __launch_bounds__(kBlock) void k_twin_q_loss(const float* __restrict__ q0, const float* __restrict__ q1,
                                                        const float* __restrict__ target, float* __restrict__ dq0,
                                                        float* __restrict__ dq1, double* __restrict__ partial, int M, float scale)
{
    __shared__ double red[kBlock / 64];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    double sq = 0.0;
    if (i < M) {
        const float a = q0[i], b = q1[i];
        const bool first = a <= b;
        const float diff = (first ? a : b) - target[i];
        sq = (double)diff * (double)diff;
        const float g = 2.0f * diff * scale;
        dq0[i] = first ? g : 0.0f;
        dq1[i] = first ? 0.0f : g;
    }
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) s += red[w];
        partial[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(64) void k_twin_q_fold(const double* __restrict__ partial, int nblk, float* __restrict__ loss_out,
                                                    double inv_m)
{
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += partial[b];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (threadIdx.x == 0) loss_out[0] = (float)(s * inv_m);
}

// the same sum over many partials (the fused critic step leaves one per wave of 32 rows: 16 384 at the SHAC horizon buffer), fixed order
__global__ __launch_bounds__(1024) void k_twin_q_fold_wide(const double* __restrict__ partial, int nblk, float* __restrict__ loss_out,
                                                           double inv_m)
{
    __shared__ double red[16];
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 1024) s += partial[b];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        loss_out[0] = (float)(t * inv_m);
    }
}

__global__ __launch_bounds__(kBlock) void k_polyak(float* __restrict__ target, const float* __restrict__ param, long long n,
                                                   float one_minus_tau, float tau)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) target[i] = __fmaf_rn(tau, param[i], target[i] * one_minus_tau);   // mul_(1 - tau); add(param, alpha=tau)
}

// test hook: every CU's LDS is filled with quiet NaNs, so that a kernel which multiplies LDS columns it never wrote by zero
// weights (0 x NaN = NaN, which a ReLU then hides as 0) fails loudly in the tests instead of once in a hundred runs
__global__ __launch_bounds__(kBlock) void k_poison_lds(int floats)
{
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < floats; i += kBlock) sm[i] = __int_as_float(0x7fc00000);
    __syncthreads();
    if (sm[(threadIdx.x * 997) % floats] == 0.0f) __builtin_trap();      // keep the stores alive
}

}  // namespace vf

extern "C" {

int vf_debug_poison_lds(vf_stream_t stream)
{
    const int bytes = 64 * 1024;                                         // the default dynamic-LDS limit; 4 blocks cover a CU
    hipLaunchKernelGGL(vf::k_poison_lds, dim3(256 * 8), dim3(vf::kBlock), bytes, vf::as_stream(stream), bytes / 4);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_shac_head_fwd(const float* mu, const float* log_std, const float* eps, float* action, int32_t N, float log_std_min,
                     float log_std_max, vf_stream_t stream)
{
    if (!mu || !log_std || !eps || !action || N <= 0) return vf::fail(VF_EINVAL, "vf_shac_head_fwd: bad argument");
    hipLaunchKernelGGL(vf::k_shac_head_fwd, dim3(vf::blocks_for(N)), dim3(vf::kBlock), 0, vf::as_stream(stream),
                       reinterpret_cast<const float4*>(mu), reinterpret_cast<const float4*>(log_std),
                       reinterpret_cast<const float4*>(eps), reinterpret_cast<float4*>(action), N, log_std_min, log_std_max);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_shac_head_bwd(const float* d_action, const float* action, const float* log_std, const float* eps, float* d_mu,
                     float* d_log_std, int32_t N, float log_std_min, float log_std_max, vf_stream_t stream)
{
    if (!d_action || !action || !log_std || !eps || !d_mu || !d_log_std || N <= 0)
        return vf::fail(VF_EINVAL, "vf_shac_head_bwd: bad argument");
    hipLaunchKernelGGL(vf::k_shac_head_bwd, dim3(vf::blocks_for(N)), dim3(vf::kBlock), 0, vf::as_stream(stream),
                       reinterpret_cast<const float4*>(d_action), reinterpret_cast<const float4*>(action),
                       reinterpret_cast<const float4*>(log_std), reinterpret_cast<const float4*>(eps),
                       reinterpret_cast<float4*>(d_mu), reinterpret_cast<float4*>(d_log_std), N, log_std_min, log_std_max);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_shac_accumulate(const float* reward, const uint8_t* done, const uint8_t* ep_flags, const float* q0, const float* q1,
                       float* disc, float* loss, float* d_reward, float* next_value_row, uint8_t* ep_done_row, float gamma,
                       float scale, int32_t last_step, int32_t N, vf_stream_t stream)
{
    if (!reward || !done || !ep_flags || !q0 || !q1 || !disc || !loss || !d_reward || !next_value_row || !ep_done_row || N <= 0)
        return vf::fail(VF_EINVAL, "vf_shac_accumulate: bad argument");
    hipLaunchKernelGGL(vf::k_shac_accumulate, dim3(vf::blocks_for(N)), dim3(vf::kBlock), 0, vf::as_stream(stream), reward, done,
                       ep_flags, q0, q1, disc, loss, d_reward, next_value_row, ep_done_row, gamma, scale, last_step, N);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_shac_accumulate_horizon(const float* reward, const uint8_t* done, const uint8_t* ep_flags, const float* q0, const float* q1,
                               float* disc, float* loss, float* d_reward, float* next_value, uint8_t* ep_done, float gamma, float scale,
                               int32_t H, int32_t N, vf_stream_t stream)
{
    if (!reward || !done || !ep_flags || !q0 || !q1 || !disc || !loss || !d_reward || !next_value || !ep_done || N <= 0 || H <= 0)
        return vf::fail(VF_EINVAL, "vf_shac_accumulate_horizon: bad argument");
    hipLaunchKernelGGL(vf::k_shac_accumulate_horizon, dim3(vf::blocks_for(N)), dim3(vf::kBlock), 0, vf::as_stream(stream), reward, done,
                       ep_flags, q0, q1, disc, loss, d_reward, next_value, ep_done, gamma, scale, H, N);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int64_t vf_twin_q_loss_scratch_doubles(int32_t M) { return M > 0 ? vf::blocks_for(M) : 0; }

int vf_twin_q_loss(const float* q0, const float* q1, const float* target, float* dq0, float* dq1, float* loss_out,
                   double* scratch, int32_t M, int64_t M_global, vf_stream_t stream)
{
    if (!q0 || !q1 || !target || !dq0 || !dq1 || !loss_out || !scratch || M <= 0 || M_global < M)
        return vf::fail(VF_EINVAL, "vf_twin_q_loss: bad argument");
    const int nb = vf::blocks_for(M);
    hipLaunchKernelGGL(vf::k_twin_q_loss, dim3(nb), dim3(vf::kBlock), 0, vf::as_stream(stream), q0, q1, target, dq0, dq1, scratch, M,
                       (float)(1.0 / (double)M_global));
    hipLaunchKernelGGL(vf::k_twin_q_fold, dim3(1), dim3(64), 0, vf::as_stream(stream), scratch, nb, loss_out, 1.0 / (double)M_global);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int64_t vf_twin_q_update_scratch_doubles(int32_t M) { return M > 0 ? (M + 31) / 32 : 0; }

int vf_twin_q_update(const vf_mlp_desc* fwd, const vf_mlp_bwd_desc* bwd, const float* params, const float* packed, const float* in0,
                     const float* in1, const float* target, float* loss_out, double* scratch, int32_t M, int64_t M_global,
                     vf_stream_t stream)
{
    if (!fwd || !bwd || !params || !packed || !in0 || !in1 || !target || !loss_out || !scratch || M <= 0 || M_global < M)
        return vf::fail(VF_EINVAL, "vf_twin_q_update: bad argument");
    if (fwd->n_layers < 1 || fwd->n_layers > VF_MLP_MAX_LAYERS || bwd->n_layers < 1 || bwd->n_layers > VF_MLP_MAX_LAYERS)
        return vf::fail(VF_EINVAL, "vf_twin_q_update: bad layer count");
    hipStream_t st = vf::as_stream(stream);
    const int rc = vf::twin_q_update_chain_try(fwd, bwd, params, packed, in0, in1, target, scratch, (float)(1.0 / (double)M_global), M, st);
    if (rc < 0) return rc;
    if (rc == 0) return vf::fail(VF_EUNSUPPORTED, "vf_twin_q_update: the layer tables are not the instantiated twin-critic class");
    hipLaunchKernelGGL(vf::k_twin_q_fold_wide, dim3(1), dim3(1024), 0, st, scratch, (M + 31) / 32, loss_out, 1.0 / (double)M_global);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_polyak_update(float* target, const float* param, int64_t n, double tau, vf_stream_t stream)
{
    if (!target || !param || n <= 0) return vf::fail(VF_EINVAL, "vf_polyak_update: bad argument");
    hipLaunchKernelGGL(vf::k_polyak, dim3((unsigned)((n + vf::kBlock - 1) / vf::kBlock)), dim3(vf::kBlock), 0, vf::as_stream(stream),
                       target, param, (long long)n, (float)(1.0 - tau), (float)tau);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

}  // extern "C"
