// vf_bptt_reverse_sac.hip -- k_bptt_reverse for the reference's own actor (utils/policies/td_policies.py:146-252): per step the adjoint
// of the env step, then k_shac_head_bwd's arithmetic in the reverse chain's head prologue (d_mu = d_a (1 - a^2), d_log_std = d_mu eps
// exp(log_std) inside the clamp interval, from the log_std rows the forward launch saved) and BOTH trunks of the 16-row reverse chain
// down to the observation gradient.  16 agents per wave for N <= 16 384 per launch, 32 above (vf_mlp_backward_data's choice for N rows; r05).  Same kernel template as the MlpPolicy classes' (vf_bptt_reverse_kernel.hpp).
#include "vf_bptt_reverse_kernel.hpp"

namespace vf {

RevKernel pick_rev_sac(int net, int kind, const vf_dyn_cfg& c, bool ckpt)
{
    if (net == 3 && kind == VF_ENV_HOVER) return pick_rev<NetSacHover, 16, VF_ENV_HOVER>(c, ckpt);
    if (net == 3 && kind == VF_ENV_RACING) return pick_rev<NetSacHover, 16, VF_ENV_RACING>(c, ckpt);
    if (net == 4 && kind == VF_ENV_NAV) return pick_rev<NetSacNav, 16, VF_ENV_NAV>(c, ckpt);
    return nullptr;
}

// ... and with 32 rows per wave (N > 16 384 per launch: vf_mlp_backward_data's choice for N rows), r05
RevKernel pick_rev_sac32(int net, int kind, const vf_dyn_cfg& c)
{
    if (net == 3 && kind == VF_ENV_HOVER) return pick_rev<NetSacHover, 32, VF_ENV_HOVER>(c, false);
    if (net == 3 && kind == VF_ENV_RACING) return pick_rev<NetSacHover, 32, VF_ENV_RACING>(c, false);
    if (net == 4 && kind == VF_ENV_NAV) return pick_rev<NetSacNav, 32, VF_ENV_NAV>(c, false);
    return nullptr;
}

}  // namespace vf
