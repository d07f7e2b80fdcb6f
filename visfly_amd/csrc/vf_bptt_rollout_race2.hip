// vf_bptt_rollout_race2.hip -- k_bptt_rollout for RacingEnv2 (envs/RacingEnv.py:218-267): RacingEnv's step under a policy that reads the
// 16-column gate-relative row -- the kernel-side env kind VF_ENV_RACING2 (vf_env_device.hpp: race2_obs; the epilogue forms the row of the
// agent's CURRENT gate, the slots / obs_final are 16 wide).  The one-observation classes (policy trunk, or the reference's two-trunk Actor),
// the motor-lag form of the interval (the reference's default).  Same kernel template (vf_bptt_rollout_kernel.hpp); r06.
#include "vf_bptt_rollout_kernel.hpp"

namespace vf {

RollKernel pick_roll_race2(int cls, const vf_dyn_cfg& c)
{
    if (!c.ctrl_delay) return nullptr;
    if (cls == 1) return pick_roll<NetHoverPi, VF_ENV_RACING2, true>(c);
    if (cls == 3) return pick_roll<NetSacHover, VF_ENV_RACING2, true>(c);
    return nullptr;
}

}  // namespace vf
