// vf_mlp_chain_sac.hip -- register-chained kernels for the reference's SAC-style Actor (utils/policies/td_policies.py:146-252):
// features extractor -> two trunks of one shape, latent_pi -> mu (4) and log_latent_pi -> log_std (4), i.e. the actor-critic class
// with a 4-wide second head (ChainNet<.., HV = 4>).  It is the actor of the reference's BPTT (utils/algorithms/BPTT.py:113) and SHAC
// (utils/algorithms/shac.py:219) loops; until r04 it ran on the block-tile kernels (k_mlp_forward / k_mlp_backward).
//   forward            k_mlp_forward_chain / _chain16 <NetSacHover | NetSacNav>: mu -> out0 (M,4), log_std -> out1 (M,4)
//   reverse            k_mlp_backward_chain <BwdProg<Net, pi, vf, IG>> from d_mu / d_log_std (M,4): both trunks, with and without the
//                      observation gradient (the first step of a horizon does not need it); masked layer gradients for k_mlp_wgrad
// ... and for its twin ContinuousCritic (:82-143): the critic's own extractor, th.cat([features, actions]) (:137) -> qf0 / qf1 -> Q, i.e.
// the class with a pass-through input and two 1-wide heads (ChainNet<.., HV = 1, HM = 1, PASS = 1>): the action columns are one more
// input tile of the trunks' first layers (68 = 64 + 4 wide), the layer table's frozen identity layer (vf_mlp_desc.identity_mask) is
// not executed.  5 critic updates over the H N rows of the horizon buffer are 2/3 of a SHAC iteration.
//   forward            <NetCriticHover>: Q1 -> out0 (M,), Q2 -> out1 (M,); inputs: the extractor's observation, then the action
//   reverse            BwdProg<Net, pi, vf, no observation gradient> from dQ1 / dQ2
// A translation unit of its own so that its instances compile next to vf_mlp_chain.hip's, not after them.
#include "vf_chain_plugin.hpp"

namespace vf {

static bool sac_off()
{
    static const bool off = [] { const char* e = getenv("VISFLY_AMD_MLP_CHAIN"); return e && atoi(e) == 0; }();
    return off;
}

// 1: launched, 0: not one of the SAC actor classes (or no second output), < 0: error
int mlp_forward_chain_try_sac(const vf_mlp_desc* d, const float* params, const float* packed, const float* in0, const float* in1,
                              const float* in2, float* out0, float* out1, int M, hipStream_t st, int M_choice)
{
    if (sac_off() || !out0 || !out1) return 0;
    const ReparamFwd rp{};
    // twin critic: 1-wide heads (no alignment requirement on the outputs); the action is the input behind the extractor's
    // (over a StateTarget extractor the concatenation would be 132 wide: beyond the 128 columns every layer kernel here is built for)
    if (chain_matches<NetCriticHover>(*d) && in1) return chain_launch<NetCriticHover>(*d, params, packed, in0, in1, out0, out1, M, st, rp, nullptr, M_choice);
    if ((reinterpret_cast<uintptr_t>(out0) | reinterpret_cast<uintptr_t>(out1)) & 15) return 0;
    if (chain_matches<NetSacNav>(*d) && in1) return chain_launch<NetSacNav>(*d, params, packed, in0, in1, out0, out1, M, st, rp, nullptr, M_choice);
    if (chain_matches<NetSacHover>(*d)) return chain_launch<NetSacHover>(*d, params, packed, in0, nullptr, out0, out1, M, st, rp, nullptr, M_choice);
    return 0;
}

// packed == nullptr: capability query only
int mlp_backward_chain_try_sac(const vf_mlp_bwd_desc* d, const float* packed, int M, hipStream_t st)
{
    if (sac_off()) return 0;
    const bool launch = packed != nullptr;
    const ReparamBwd rp{};
    if (bwd_chain_matches<NetSacNav, true, true, true>(*d)) return launch ? bwd_chain_launch<NetSacNav, true, true, true>(*d, packed, M, st, rp) : 1;
    if (bwd_chain_matches<NetSacNav, true, true, false>(*d)) return launch ? bwd_chain_launch<NetSacNav, true, true, false>(*d, packed, M, st, rp) : 1;
    if (bwd_chain_matches<NetSacHover, true, true, true>(*d)) return launch ? bwd_chain_launch<NetSacHover, true, true, true>(*d, packed, M, st, rp) : 1;
    if (bwd_chain_matches<NetSacHover, true, true, false>(*d)) return launch ? bwd_chain_launch<NetSacHover, true, true, false>(*d, packed, M, st, rp) : 1;
    if (bwd_chain_matches<NetCriticHover, true, true, false>(*d)) return launch ? bwd_chain_launch<NetCriticHover, true, true, false>(*d, packed, M, st, rp) : 1;
    return 0;
}

// (the fused critic step k_twin_q_update_chain: vf_mlp_chain_kernels.hpp)

// fused critic step (forward + twin-Q loss + reverse chain): 1 launched, 0 not the instantiated class, < 0 error.  part: ceil(M / 32) doubles
int twin_q_update_chain_try(const vf_mlp_desc* d, const vf_mlp_bwd_desc* bd, const float* params, const float* packed, const float* in0,
                            const float* in1, const float* target, double* part, float scale, int M, hipStream_t st)
{
    if (sac_off()) return 0;
    for (int i = 0; i < d->n_layers; ++i)
        if (d->layer[i].dst < VF_MLP_OUT0 && !d->layer[i].save && !((d->identity_mask >> i) & 1)) return 0;   // the weight gradients need every layer input
    for (int i = 0; i < d->n_layers; ++i)
        if (d->layer[i].save && !rows_fit_u32(M, d->layer[i].save_ld)) return 0;
    for (int l = 0; l < bd->n_layers; ++l)
        if (!rows_fit_u32(M, bd->layer[l].ld_dy)) return 0;
    ChainArgs g{*d, params, packed, ChainIo{{in0, in1}, nullptr, nullptr}, M, nullptr, nullptr, nullptr, {nullptr, nullptr}};
    BwdArgsChain gb{*bd, packed, M, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (!(chain_matches<NetCriticHover>(*d) && in1 && bwd_chain_matches<NetCriticHover, true, true, false>(*bd))) {
        for (int i = 0; i < chain_plugin_count(); ++i)      // a twin critic of another shape: its generated class (vf_mlp_chain_gen.hpp, PASS = 1)
            if (chain_plugin(i)->twin_q_update)
                if (int rc = chain_plugin(i)->twin_q_update(&g, &gb, target, part, scale, M, st)) {
                    if (rc == 1) chain_plugin_count_launch();
                    if (rc <= -1000) return fail(VF_EHIP, "vf_twin_q_update (chain plugin) failed: %s", hipGetErrorString((hipError_t)(-rc - 1000)));
                    return rc;
                }
        return 0;
    }
    const int sp = twin_q_update_split_try(g, gb, target, part, scale, M, st);     // two waves per row tile (vf_mlp_chain_split.hip)
    if (sp) return sp;
    hipLaunchKernelGGL(k_twin_q_update_chain<NetCriticHover>, dim3((M + 31) / 32), dim3(64), 0, st, g, gb, target, part, scale);
    VF_HIP(hipGetLastError());
    return 1;
}

}  // namespace vf
