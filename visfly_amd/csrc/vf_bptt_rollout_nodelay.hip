// vf_bptt_rollout_nodelay.hip -- k_bptt_rollout for dynamics WITHOUT the motor lag (ctrl_delay = False: the rotors take their set
// points at once, envs/base/dynamics.py:534-554 -- not the reference's default, none of its configurations uses it; until r05 these
// launches stepped launch by launch).  Same kernel template as every other instance (vf_bptt_rollout_kernel.hpp), all four actor
// classes; a translation unit of its own so that the instances compile side by side.
#include "vf_bptt_rollout_kernel.hpp"

namespace vf {

RollKernel pick_roll_nodelay(int cls, int kind, const vf_dyn_cfg& c)
{
    if (cls == 1 && kind == VF_ENV_HOVER) return pick_roll<NetHoverPi, VF_ENV_HOVER, false>(c);
    if (cls == 1 && kind == VF_ENV_RACING) return pick_roll<NetHoverPi, VF_ENV_RACING, false>(c);
    if (cls == 2 && kind == VF_ENV_NAV) return pick_roll<NetNavPi, VF_ENV_NAV, false>(c);
    if (cls == 3 && kind == VF_ENV_HOVER) return pick_roll<NetSacHover, VF_ENV_HOVER, false>(c);
    if (cls == 3 && kind == VF_ENV_RACING) return pick_roll<NetSacHover, VF_ENV_RACING, false>(c);
    if (cls == 4 && kind == VF_ENV_NAV) return pick_roll<NetSacNav, VF_ENV_NAV, false>(c);
    return nullptr;
}

}  // namespace vf
