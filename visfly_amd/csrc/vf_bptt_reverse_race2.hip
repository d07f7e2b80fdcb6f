// vf_bptt_reverse_race2.hip -- k_bptt_reverse for RacingEnv2: the one-observation network classes over the kernel-side env kind
// VF_ENV_RACING2 (see vf_bptt_rollout_race2.hip); the 16-column observation's adjoint is race2_obs_bwd in front of the raw row's
// (vf_env_bwd_body.hpp), g_obs rows are 16 wide.  Same kernel template (vf_bptt_reverse_kernel.hpp); r06.
#include "vf_bptt_reverse_kernel.hpp"

namespace vf {

RevKernel pick_rev_race2(int net, bool r16, const vf_dyn_cfg& c, bool ckpt)
{
    if (!c.ctrl_delay) return nullptr;
    if (net == 1) return r16 ? pick_rev<NetHover, 16, VF_ENV_RACING2, true>(c, ckpt) : pick_rev<NetHover, 32, VF_ENV_RACING2, true>(c, false);
    if (net == 3) return r16 ? pick_rev<NetSacHover, 16, VF_ENV_RACING2, true>(c, ckpt) : pick_rev<NetSacHover, 32, VF_ENV_RACING2, true>(c, false);
    return nullptr;
}

}  // namespace vf
