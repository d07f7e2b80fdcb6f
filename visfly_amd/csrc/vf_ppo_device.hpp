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

// Squashed diagonal Gaussian head for row i (policies.py:177-181, SB3 SquashedDiagGaussianDistribution): action = tanh(mean +
// exp(log_std) eps), eps = Box-Muller over Philox(row, step; seed); -> log-prob of the action.  k_head_sample (vf_ppo.hip) and
// the persistent PPO roll-out (vf_bptt_rollout.hip) share it.
__device__ __forceinline__ float head_sample_row(const float4 m4, const float* __restrict__ log_std, int i, unsigned long long seed,
                                                 unsigned long long step, int deterministic, float4& action)
{
    const float mu[4] = {m4.x, m4.y, m4.z, m4.w};
    const float ls[4] = {log_std[0], log_std[1], log_std[2], log_std[3]};
    float a[4];
    if (deterministic) {
#pragma unroll
        for (int d = 0; d < 4; ++d) a[d] = tanhf(mu[d]);
    } else {
        const U4 r = philox4x32_10(U4{(unsigned)i, (unsigned)step, (unsigned)(step >> 32), 0xac7u}, (unsigned)seed,
                                   (unsigned)(seed >> 32));
        const float u1 = ((float)(r.x >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(r.y >> 8) * (1.0f / 16777216.0f);
        const float u3 = ((float)(r.z >> 8) + 1.0f) * (1.0f / 16777216.0f), u4 = (float)(r.w >> 8) * (1.0f / 16777216.0f);
        const float ra = sqrtf(-2.0f * logf(u1)), rb = sqrtf(-2.0f * logf(u3));
        const float two_pi = 6.28318530717958647692f;
        const float e[4] = {ra * cosf(two_pi * u2), ra * sinf(two_pi * u2), rb * cosf(two_pi * u4), rb * sinf(two_pi * u4)};
#pragma unroll
        for (int d = 0; d < 4; ++d) a[d] = tanhf(mu[d] + expf(ls[d]) * e[d]);
    }
    float g[4];
    const float lp = squashed_log_prob(mu, ls, a, g);
    action = make_float4(a[0], a[1], a[2], a[3]);
    return lp;
}

// The part of a row's loss that needs the ACTION only (SquashedDiagGaussianDistribution.log_prob: gaussian_actions = atanh(clamp(a)),
// the tanh correction log(1 - a^2 + 1e-6)): 12 of the row's ~26 transcendental evaluations.  The fused
// update kernels run it BEFORE the forward chain, while the wave waits for its first weight fragments anyway, instead of between the
// forward and the reverse chain where the matrix pipe idles for it (r05: 8.9 k of a wave's 136 k cycles, profiles/r05_chain_split.txt).
struct PpoRowPre {
    float g[4], corr[4];
};

__device__ __forceinline__ PpoRowPre ppo_row_pre(const float* a)
{
    PpoRowPre p;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        p.g[d] = atanh_clamped(a[d]);
        p.corr[d] = logf(1.0f - a[d] * a[d] + 1e-6f);
    }
    return p;
}

// one row of the clipped-surrogate loss: gradients w.r.t. the head outputs and the 9 statistics
// st = {policy loss, value loss, log prob, approx kl, clipped?, d_log_std[4]}
// `row`: index of this row in the call's arrays (for cfg.old_value)
// (the same operations in the same order as squashed_log_prob + the former single-piece ppo_row: same bits)
__device__ __forceinline__ void ppo_row_post(const PpoRowPre& p, const float* mu, float v, const float* ls, float old_lp, float A, float R,
                                             const vf_ppo_loss_cfg& cfg, float* dm, float& d_value, float* st, int row = 0)
{
    float lp = 0.0f, sd[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        sd[d] = expf(ls[d]);            // (wave-uniform: not worth four registers across the forward chain)
        const float z = (p.g[d] - mu[d]) / sd[d];
        lp += -0.5f * z * z - ls[d] - 0.91893853320467274178f;
        lp -= p.corr[d];
    }
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
        const float z = (p.g[d] - mu[d]) / sd[d];
        dm[d] = dl_dlp * z / sd[d];
        st[5 + d] = dl_dlp * (z * z - 1.0f);
    }
    d_value = cfg.vf_coef * 2.0f * dv * cfg.inv_batch * vgate;
    st[0] = -fminf(s1, s2);
    st[1] = dv * dv;
    st[2] = lp;
    st[3] = (ratio - 1.0f) - log_ratio;
    st[4] = fabsf(ratio - 1.0f) > cfg.clip_range ? 1.0f : 0.0f;
}

__device__ __forceinline__ void ppo_row(const float* mu, float v, const float* ls, const float* a, float old_lp, float A, float R,
                                        const vf_ppo_loss_cfg& cfg, float* dm, float& d_value, float* st, int row = 0)
{
    const PpoRowPre p = ppo_row_pre(a);
    ppo_row_post(p, mu, v, ls, old_lp, A, R, cfg, dm, d_value, st, row);
}

}  // namespace vf
