// vf_bptt_rollout_nav2.hip -- k_bptt_rollout for NavigationEnv2 (envs/NavigationEnv.py:102-224): the Navigation env kind under a policy
// with ONE observation input -- its "state" row already carries target - p -- i.e. the StateExtractor classes (policy trunk, or the
// reference's two-trunk Actor) over VF_ENV_NAV.  The observation / reward variants themselves are the epilogue's (obs_variant,
// nav2_reward).  Same kernel template (vf_bptt_rollout_kernel.hpp); r05.
#include "vf_bptt_rollout_kernel.hpp"

namespace vf {

RollKernel pick_roll_nav2(int cls, const vf_dyn_cfg& c)
{
    if (cls == 1) return c.ctrl_delay ? pick_roll<NetHoverPi, VF_ENV_NAV, true>(c) : pick_roll<NetHoverPi, VF_ENV_NAV, false>(c);
    if (cls == 3) return c.ctrl_delay ? pick_roll<NetSacHover, VF_ENV_NAV, true>(c) : pick_roll<NetSacHover, VF_ENV_NAV, false>(c);
    return nullptr;
}

}  // namespace vf
