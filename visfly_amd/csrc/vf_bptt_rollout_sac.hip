// vf_bptt_rollout_sac.hip -- k_bptt_rollout for the reference's own actor (utils/policies/td_policies.py:146-252, what
// BPTT.learn / SHAC's roll-out call per step: utils/algorithms/BPTT.py:113, shac.py:219): both trunks of the 16-row chain run per
// step -- latent_pi -> mu, log_latent_pi -> log_std -- and the action head is k_shac_head_fwd's arithmetic on the two heads still in
// registers, a = tanh(mu + eps exp(clamp(log_std, -10, 2))) (chain16_epilogue, HV == 4).  The log_std rows of every step are kept
// (the head's reverse reads them), like every layer's activations.  Same kernel template as the MlpPolicy classes'
// (vf_bptt_rollout_kernel.hpp); a translation unit of its own so that the instances compile side by side.
#include "vf_bptt_rollout_kernel.hpp"

namespace vf {

RollKernel pick_roll_sac(int net, int kind, const vf_dyn_cfg& c)
{
    if (net == 3 && kind == VF_ENV_HOVER) return pick_roll<NetSacHover, VF_ENV_HOVER>(c);
    if (net == 3 && kind == VF_ENV_RACING) return pick_roll<NetSacHover, VF_ENV_RACING>(c);
    if (net == 4 && kind == VF_ENV_NAV) return pick_roll<NetSacNav, VF_ENV_NAV>(c);
    return nullptr;
}

}  // namespace vf
