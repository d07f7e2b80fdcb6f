// clip_grad_norm_ + torch.optim.Adam (weight decay as L2) on one parameter: the arithmetic shared by k_adam (vf_ppo.hip) and the
// fused tail of the weight-gradient launch (vf_mlp_wgrad.hip) -- the same IEEE operations in the same order, so the two paths
// agree to the bit.  Reference: utils/algorithms/PPO.py:285-292 (clip_grad_norm_, optimizer.step()).
#pragma once
#include <cmath>

#include "vf_common.hpp"

namespace vf {

// bias corrections of step t (host): bc1 = 1 - beta1^t, bc2_sqrt = sqrt(1 - beta2^t), rounded as torch.optim.Adam's fp32 path
inline void adam_bias(const vf_adam_cfg& c, float* bc1, float* bc2_sqrt)
{
    *bc1 = 1.0f - (float)pow((double)c.beta1, (double)c.step);
    *bc2_sqrt = sqrtf(1.0f - (float)pow((double)c.beta2, (double)c.step));
}

#ifdef __HIPCC__

// total norm from the squared norm: torch.nn.utils.clip_grad_norm_'s clip coefficient
__device__ __forceinline__ float adam_clip_coef(float sumsq, float max_grad_norm)
{
    const float total = sqrtf(sumsq);
    return fminf(max_grad_norm / (total + 1e-6f), 1.0f);
}

// -> the new parameter; m / v updated in place.  step = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t)
__device__ __forceinline__ float adam_param(float pi, float g, float& m, float& v, float coef, const vf_adam_cfg& c, float step, float bc2_sqrt)
{
    float gi = g * coef;
    gi = gi + c.weight_decay * pi;
    const float mi = c.beta1 * m + (1.0f - c.beta1) * gi;
    const float vi = c.beta2 * v + (1.0f - c.beta2) * gi * gi;
    m = mi;
    v = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + c.eps;
    return pi - step * (mi / denom);
}

// keep the packed MFMA images of the weights current (vf_mlp_pack_weights layout): four int32 offsets per parameter, -1 = none
__device__ __forceinline__ void adam_refresh_packed(const vf_adam_cfg& c, long i, float pn)
{
    const int4 o = reinterpret_cast<const int4*>(c.pack_map)[i];
    if (o.x >= 0) c.packed[o.x] = pn;
    if (o.y >= 0) c.packed[o.y] = pn;
    if (o.z >= 0) c.packed[o.z] = pn;
    if (o.w >= 0) c.packed[o.w] = pn;
}

#endif

}  // namespace vf
