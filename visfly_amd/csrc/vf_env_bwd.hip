// vf_env_bwd.hip -- adjoint of the fused env step (gfx950), for first-order policy optimisation.
//
// The reference builds a torch autograd tape of ~3.8k nodes per control step when
// requires_grad=True (utils/algorithms/BPTT.py:107-134, envs/base/dynamics.py:176-190); here the
// reverse pass of one control interval is ONE launch: the thread re-plays the interval from the
// checkpointed pre-step state (tape), parks each sub-step's inputs in LDS, then walks the
// sub-steps backwards with hand-derived adjoints of every op in SURVEY App. A.
//
// Conventions: clamp passes the gradient on the closed interval, abs'(0) = 0,
// quaternion products are differentiated as the polynomial expressions the reference evaluates
// (d(a*b): lam_a = lam * conj(b), lam_b = conj(a) * lam), comparisons / done masks carry no gradient,
// agents reset at the end of a step stop the gradient there.
#include "vf_env_bwd_body.hpp"

namespace vf {

template <int KIND, int ACT, int INTEG, bool CTRL_DELAY>
__global__ __launch_bounds__(kBlock) void k_env_step_bwd(const vf_dyn_cfg* __restrict__ cp, const vf_env_cfg* __restrict__ ep, const BwdArgs g)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [S*kSave][kBlock]
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= ((g.N + kBlock - 1) / kBlock) * kBlock) return;
    env_step_bwd_agent<KIND, ACT, INTEG, CTRL_DELAY, kBlock>(*cp, *ep, g, i, i < g.N, lds + threadIdx.x);
}

}  // namespace vf

namespace {

using BwdKernel = void (*)(const vf_dyn_cfg*, const vf_env_cfg*, const vf::BwdArgs);

template <int KIND>
BwdKernel pick_bwd(const vf_dyn_cfg& c)
{
    const int key = (c.integrator == VF_INT_RK4 ? 4 : 0) | (c.action_type == VF_ACT_BODYRATE ? 2 : 0) | (c.ctrl_delay ? 1 : 0);
    switch (key) {
    case 0: return vf::k_env_step_bwd<KIND, VF_ACT_THRUST, VF_INT_EULER, false>;
    case 1: return vf::k_env_step_bwd<KIND, VF_ACT_THRUST, VF_INT_EULER, true>;
    case 2: return vf::k_env_step_bwd<KIND, VF_ACT_BODYRATE, VF_INT_EULER, false>;
    case 3: return vf::k_env_step_bwd<KIND, VF_ACT_BODYRATE, VF_INT_EULER, true>;
    case 4: return vf::k_env_step_bwd<KIND, VF_ACT_THRUST, VF_INT_RK4, false>;
    case 5: return vf::k_env_step_bwd<KIND, VF_ACT_THRUST, VF_INT_RK4, true>;
    case 6: return vf::k_env_step_bwd<KIND, VF_ACT_BODYRATE, VF_INT_RK4, false>;
    default: return vf::k_env_step_bwd<KIND, VF_ACT_BODYRATE, VF_INT_RK4, true>;
    }
}

}  // namespace

extern "C" int vf_env_step_bwd(vf_env* h, const vf_env_bwd_args* a, vf_stream_t stream)
{
    if (!h || !a || !a->tape_slab || !a->action || !a->done || !a->adj_slab || !a->d_action)
        return vf::fail(VF_EINVAL, "vf_env_step_bwd: null argument");
    if (h->dyn.cfg.action_type != VF_ACT_THRUST && h->dyn.cfg.action_type != VF_ACT_BODYRATE)
        return vf::fail(VF_EINVAL, "vf_env_step_bwd: the adjoint covers the thrust and bodyrate action types only");
    if (h->dyn.cfg.integrator != VF_INT_EULER && h->dyn.cfg.integrator != VF_INT_RK4)
        return vf::fail(VF_EINVAL, "vf_env_step_bwd: unknown integrator");
    if (h->dyn.wind)
        return vf::fail(VF_EUNSUPPORTED, "vf_env_step_bwd: per-agent wind rows are set (vf_dyn_set_wind); the adjoint replays the "
                                         "interval with the constant vf_dyn_cfg.wind and would differentiate another trajectory");
    const int S = h->dyn.cfg.interval_steps;
    if (S > 10) return vf::fail(VF_EINVAL, "vf_env_step_bwd: at most 10 sub-steps per control interval (LDS budget)");
    const size_t lds = (size_t)S * vf::kSave * vf::kBlock * sizeof(float);
    BwdKernel k = h->cfg.kind == VF_ENV_RACING ? pick_bwd<VF_ENV_RACING>(h->dyn.cfg)
                  : (h->cfg.kind == VF_ENV_NAV ? pick_bwd<VF_ENV_NAV>(h->dyn.cfg) : pick_bwd<VF_ENV_HOVER>(h->dyn.cfg));
    if (lds > 64 * 1024)
        VF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    vf::BwdArgs g{h->dyn.N, h->dyn.G, h->dyn.g_drag, h->g_race, a->tape_slab, reinterpret_cast<const float4*>(a->action),
                  a->d_obs, a->d_reward, a->done, a->adj_slab, reinterpret_cast<float4*>(a->d_action)};
    hipLaunchKernelGGL(k, dim3(h->dyn.Npad / vf::kBlock), dim3(vf::kBlock), lds, vf::as_stream(stream), h->dyn.d_cfg, h->d_cfg, g);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

// Dynamics.step's own reverse pass (SURVEY 8b.4 `vf_dyn_step_bwd`): what autograd does for a loss on the states a bare
// `Dynamics` object returns (no env layer, hence no reward term and no episode ends) -- the env adjoint above with d_reward = NULL
// and nobody done, on a vf_dyn handle.  The kernel wants an env constant block: a zeroed one, uploaded once per handle.
extern "C" int vf_dyn_step_bwd(vf_dyn* h, const float* tape_slab, const float* action, const float* d_state, float* adj_slab,
                               float* d_action, vf_stream_t stream)
{
    if (!h || !tape_slab || !action || !adj_slab || !d_action) return vf::fail(VF_EINVAL, "vf_dyn_step_bwd: null argument");
    if (h->cfg.action_type != VF_ACT_THRUST && h->cfg.action_type != VF_ACT_BODYRATE)
        return vf::fail(VF_EINVAL, "vf_dyn_step_bwd: the adjoint covers the thrust and bodyrate action types only");
    if (h->wind)
        return vf::fail(VF_EUNSUPPORTED, "vf_dyn_step_bwd: per-agent wind rows are set (vf_dyn_set_wind): see vf_env_step_bwd");
    const int S = h->cfg.interval_steps;
    if (S > 10) return vf::fail(VF_EINVAL, "vf_dyn_step_bwd: at most 10 sub-steps per control interval (LDS budget)");
    if (!h->d_cfg) return vf::fail(VF_ESTATE, "vf_dyn_step_bwd: the handle has no device constant block");
    if (!h->d_env_dummy) {
        vf_env_cfg z{};
        z.kind = VF_ENV_HOVER;
        if (int rc = vf::upload_cfg(z, &h->d_env_dummy)) return rc;
    }
    const size_t lds = (size_t)S * vf::kSave * vf::kBlock * sizeof(float);
    BwdKernel k = pick_bwd<VF_ENV_HOVER>(h->cfg);
    if (lds > 64 * 1024)
        VF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    vf::BwdArgs g{h->N, h->G, h->g_drag, -1, tape_slab, reinterpret_cast<const float4*>(action), d_state, nullptr, nullptr, adj_slab,
                  reinterpret_cast<float4*>(d_action)};
    hipLaunchKernelGGL(k, dim3(h->Npad / vf::kBlock), dim3(vf::kBlock), lds, vf::as_stream(stream), h->d_cfg, h->d_env_dummy, g);
    VF_HIP(hipGetLastError());
    return VF_OK;
}
