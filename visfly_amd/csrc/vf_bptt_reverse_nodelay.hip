// vf_bptt_reverse_nodelay.hip -- k_bptt_reverse for dynamics WITHOUT the motor lag (ctrl_delay = False, envs/base/dynamics.py:534-554):
// the instances of vf_bptt_reverse.hip / vf_bptt_reverse_sac.hip once more with the direct form of the interval's adjoint.  Same kernel
// template (vf_bptt_reverse_kernel.hpp); a translation unit of its own so that the instances compile side by side.
#include "vf_bptt_reverse_kernel.hpp"

namespace vf {

RevKernel pick_rev_nodelay(int net, bool r16, int kind, const vf_dyn_cfg& c, bool ckpt)
{
    if (net == 1 && kind == VF_ENV_HOVER) return r16 ? pick_rev<NetHover, 16, VF_ENV_HOVER, false>(c, ckpt) : pick_rev<NetHover, 32, VF_ENV_HOVER, false>(c, ckpt);
    if (net == 1 && kind == VF_ENV_RACING) return r16 ? pick_rev<NetHover, 16, VF_ENV_RACING, false>(c, ckpt) : pick_rev<NetHover, 32, VF_ENV_RACING, false>(c, ckpt);
    if (net == 2 && kind == VF_ENV_NAV) return r16 ? pick_rev<NetNav, 16, VF_ENV_NAV, false>(c, ckpt) : pick_rev<NetNav, 32, VF_ENV_NAV, false>(c, ckpt);
    if (net == 3 && kind == VF_ENV_HOVER) return r16 ? pick_rev<NetSacHover, 16, VF_ENV_HOVER, false>(c, ckpt) : pick_rev<NetSacHover, 32, VF_ENV_HOVER, false>(c, false);
    if (net == 3 && kind == VF_ENV_RACING) return r16 ? pick_rev<NetSacHover, 16, VF_ENV_RACING, false>(c, ckpt) : pick_rev<NetSacHover, 32, VF_ENV_RACING, false>(c, false);
    if (net == 4 && kind == VF_ENV_NAV) return r16 ? pick_rev<NetSacNav, 16, VF_ENV_NAV, false>(c, ckpt) : pick_rev<NetSacNav, 32, VF_ENV_NAV, false>(c, false);
    return nullptr;
}

}  // namespace vf
