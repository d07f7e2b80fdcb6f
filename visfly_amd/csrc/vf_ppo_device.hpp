// Per-row arithmetic of the PPO loss (PPO.py:210-263; SB3 SquashedDiagGaussianDistribution), shared by the stand-alone
// loss kernel (vf_ppo.hip) and the fused forward + loss + reverse-chain kernel (vf_mlp_chain.hip).
#pragma once
#include "vf_common.hpp"

namespace vf {

constexpr int kStats = 16;   // floats per partial row of loss statistics (9 used)

__device__ __forceinline__ float atanh_clamped(float a)
{
    // TanhBijector.inverse: atanh(clamp(a, -1 + eps, 1 - eps)), eps = float32 eps
    const float eps = 1.1920929e-07f;
    const float x = fminf(fmaxf(a, -1.0f + eps), 1.0f - eps);
    return 0.5f * (log1pf(x) - log1pf(-x));
}

// log N(g; mu, sigma) summed over 4 dims minus the tanh correction sum log(1 - a^2 + 1e-6)
__device__ __forceinline__ float squashed_log_prob(const float* mu, const float* ls, const float* a, float* g)
{
    float lp = 0.0f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        g[d] = atanh_clamped(a[d]);
        const float sd = expf(ls[d]);
        const float z = (g[d] - mu[d]) / sd;
        lp += -0.5f * z * z - ls[d] - 0.91893853320467274178f;
        lp -= logf(1.0f - a[d] * a[d] + 1e-6f);
    }
    return lp;
}

// one row of the clipped-surrogate loss: gradients w.r.t. the head outputs and the 9 statistics
// st = {policy loss, value loss, log prob, approx kl, clipped?, d_log_std[4]}
// `row`: index of this row in the call's arrays (for cfg.old_value)
__device__ __forceinline__ void ppo_row(const float* mu, float v, const float* ls, const float* a, float old_lp, float A, float R,
                                        const vf_ppo_loss_cfg& cfg, float* dm, float& d_value, float* st, int row = 0)
{
    float g[4];
    const float lp = squashed_log_prob(mu, ls, a, g);
    const float log_ratio = lp - old_lp;
    const float ratio = expf(log_ratio);
    const float lo = 1.0f - cfg.clip_range, hi = 1.0f + cfg.clip_range;
    const float rc = fminf(fmaxf(ratio, lo), hi);
    const float s1 = A * ratio, s2 = A * rc;
    const bool clipped = ratio < lo || ratio > hi;
    // d(-min(s1,s2))/d ratio: through s1 when it is the smaller one (or equal: unclipped), else 0
    const float dl_dratio = (s1 <= s2 || !clipped) ? -A : 0.0f;
    float vp = v, vgate = 1.0f;
    if (cfg.clip_range_vf > 0.0f && cfg.old_value) {   // PPO.py:237-243; torch.clamp passes the gradient inside [min, max]
        const float old_v = cfg.old_value[row], dvo = v - old_v;
        vp = old_v + fminf(fmaxf(dvo, -cfg.clip_range_vf), cfg.clip_range_vf);
        vgate = (dvo >= -cfg.clip_range_vf && dvo <= cfg.clip_range_vf) ? 1.0f : 0.0f;
    }
    const float dv = vp - R;
    // d loss / d log_prob per row (means over the global batch)
    const float dl_dlp = (dl_dratio * ratio + cfg.ent_coef) * cfg.inv_batch;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const float sd = expf(ls[d]);
        const float z = (g[d] - mu[d]) / sd;
        dm[d] = dl_dlp * z / sd;
        st[5 + d] = dl_dlp * (z * z - 1.0f);
    }
    d_value = cfg.vf_coef * 2.0f * dv * cfg.inv_batch * vgate;
    st[0] = -fminf(s1, s2);
    st[1] = dv * dv;
    st[2] = lp;
    st[3] = (ratio - 1.0f) - log_ratio;
    st[4] = fabsf(ratio - 1.0f) > cfg.clip_range ? 1.0f : 0.0f;
}

}  // namespace vf
