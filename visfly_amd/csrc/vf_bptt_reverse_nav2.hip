// vf_bptt_reverse_nav2.hip -- k_bptt_reverse for NavigationEnv2: the one-observation network classes over the Navigation env kind
// (see vf_bptt_rollout_nav2.hip); the adjoint of the observation / reward variants is env_step_bwd_agent's (obs_variant_bwd, the
// NAV2 branch of the reward gradient).  Same kernel template (vf_bptt_reverse_kernel.hpp); r05.
#include "vf_bptt_reverse_kernel.hpp"

namespace vf {

template <bool DELAY>
static RevKernel pick_nav2(int net, bool r16, const vf_dyn_cfg& c, bool ckpt)
{
    if (net == 1) return r16 ? pick_rev<NetHover, 16, VF_ENV_NAV, DELAY>(c, ckpt) : pick_rev<NetHover, 32, VF_ENV_NAV, DELAY>(c, false);
    if (net == 3) return r16 ? pick_rev<NetSacHover, 16, VF_ENV_NAV, DELAY>(c, ckpt) : pick_rev<NetSacHover, 32, VF_ENV_NAV, DELAY>(c, false);
    return nullptr;
}

RevKernel pick_rev_nav2(int net, bool r16, const vf_dyn_cfg& c, bool ckpt)
{
    return c.ctrl_delay ? pick_nav2<true>(net, r16, c, ckpt) : pick_nav2<false>(net, r16, c, ckpt);
}

}  // namespace vf
