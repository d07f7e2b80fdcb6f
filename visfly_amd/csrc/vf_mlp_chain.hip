// Register-chained actor-critic forward for gfx950: one wave64 carries 32 rows through the WHOLE network, the
// activations never leave its registers.
//
// Every layer is computed transposed, Y^T = W X^T, with v_mfma_f32_32x32x2_f32:
//     A operand = weights   A[i = n][k]      lane = n + 32 kk
//     B operand = X^T       B[k][j = m]      lane = m + 32 kk
//     C / D     = Y^T       D[i = n][j = m]  lane (m, h = lane >> 5) holds n = 4 h + (r & 3) + 8 (r >> 2), r = 0..15
// A lane of the accumulator therefore holds 16 features of ITS OWN row m -- exactly what the B operand of the next
// layer wants from that lane, provided step s of the next layer reduces over k = kidx(s, h) = the feature register
// r = s % 16 of tile s / 16 holds: kidx = 32 (s / 16) + 8 ((s % 16) / 4) + 4 h + (s % 4).  The reduction order
// is free (a sum), so the weights are packed once in that order (k_mlp_pack_weights, image at vf_mlp_layer.wr_off:
// one contiguous 1 KiB block = the A fragments of four consecutive steps of one 32-feature output tile, one float4
// per lane) and the accumulator registers are fed back as B operands untouched: no LDS, no barriers, no
// inter-wave traffic.  Bias + ReLU run on the accumulator registers; the copies the backward needs are stored from
// them (float4 = 4 consecutive features of a row).
//
// The layer shapes must be compile-time (register arrays cannot be indexed at run time): the kernel is a template
// over the network class of the reference's policies (utils/policies/extractors.py:578-592,662-678;
// policies.py:18-49): NB extractor branches of two ReLU layers, concatenated, then policy / value trunks of two
// ReLU layers with a 4-wide / 1-wide head.  vf_mlp_forward picks it when the layer table matches an instantiated
// shape and falls back to the LDS kernel (k_mlp_forward) otherwise.
#include "vf_chain_plugin.hpp"

namespace vf {

// a plugin's return value in the library's terms (vf_chain_plugin.hpp)
static int plugin_rc(int rc, const char* what, bool launched = true)
{
    if (rc == 1 && launched) chain_plugin_count_launch();      // (a capability query is not a launch)
    if (rc <= -1000) return fail(VF_EHIP, "%s (chain plugin) failed: %s", what, hipGetErrorString((hipError_t)(-rc - 1000)));
    return rc;
}

// (the 32-row forward's device code -- chain_load .. chain_prologue -- lives in vf_mlp_chain.hpp: vf_ppo_rollout.hip runs it too)

// (the fused PPO minibatch step k_ppo_update_chain: vf_mlp_chain_kernels.hpp)

int mlp_backward_chain_try(const vf_mlp_bwd_desc* d, const float* packed, int M, hipStream_t st, const ReparamBwd* rpp)
{
    for (int l = 0; l < d->n_layers; ++l)
        if (!rows_fit_u32(M, d->layer[l].ld_dy)) return 0;
    const ReparamBwd rp = rpp ? *rpp : ReparamBwd{};
    static const bool off = [] { const char* e = getenv("VISFLY_AMD_MLP_CHAIN"); return e && atoi(e) == 0; }();
    if (off) return 0;
    const bool launch = packed != nullptr;       // packed == nullptr: capability query only
    if (bwd_chain_matches<NetNav, true, true, false>(*d)) return launch ? bwd_chain_launch<NetNav, true, true, false>(*d, packed, M, st, rp) : 1;
    if (bwd_chain_matches<NetNav, true, false, true>(*d)) return launch ? bwd_chain_launch<NetNav, true, false, true>(*d, packed, M, st, rp) : 1;
    if (bwd_chain_matches<NetHover, true, true, false>(*d)) return launch ? bwd_chain_launch<NetHover, true, true, false>(*d, packed, M, st, rp) : 1;
    if (bwd_chain_matches<NetHover, true, false, true>(*d)) return launch ? bwd_chain_launch<NetHover, true, false, true>(*d, packed, M, st, rp) : 1;
    for (int i = 0; i < chain_plugin_count(); ++i)      // shapes compiled on first use (vf_mlp_chain_gen.hpp)
        if (chain_plugin(i)->backward)
            if (int rc = chain_plugin(i)->backward(d, launch ? packed : nullptr, M, st, rpp)) return plugin_rc(rc, "vf_mlp_backward_data", launch);
    if (!rpp) return mlp_backward_chain_try_sac(d, packed, M, st);       // the SAC-style Actor's classes (vf_mlp_chain_sac.hip)
    return 0;
}

// fused PPO step (forward + loss + reverse chain): 1 launched, 0 not an instantiated class, < 0 error.
// part: ceil(M / 32) x kStats floats of loss-statistic partials
int ppo_update_chain_try(const vf_mlp_desc* d, const vf_mlp_bwd_desc* bd, const float* params, const float* packed, const float* in0,
                         const float* in1, const float* log_std, const float* action, const float* old_lp, const float* adv,
                         const float* ret, float* part, const vf_ppo_loss_cfg* cfg, int M, hipStream_t st)
{
    static const bool off = [] { const char* e = getenv("VISFLY_AMD_MLP_CHAIN"); return e && atoi(e) == 0; }();
    if (off) return 0;
    for (int i = 0; i < d->n_layers; ++i)
        if (d->layer[i].dst < VF_MLP_OUT0 && !d->layer[i].save) return 0;          // the weight gradients need every layer input
    for (int i = 0; i < d->n_layers; ++i)
        if (d->layer[i].save && !rows_fit_u32(M, d->layer[i].save_ld)) return 0;
    for (int l = 0; l < bd->n_layers; ++l)
        if (!rows_fit_u32(M, bd->layer[l].ld_dy)) return 0;
    if (cfg->row_index && (!cfg->obs_copy0 || (in1 && !cfg->obs_copy1)))
        return fail(VF_EINVAL, "vf_ppo_update: row_index needs obs_copy0 / obs_copy1 (the weight gradients read the observation rows in call order)");
    ChainArgs g{*d, params, packed, ChainIo{{in0, in1}, nullptr, nullptr}, M, nullptr, nullptr, nullptr,
                {cfg->row_index ? cfg->obs_copy0 : nullptr, cfg->row_index ? cfg->obs_copy1 : nullptr}};
    BwdArgsChain gb{*bd, packed, M, nullptr, nullptr, nullptr, nullptr, nullptr};
    PpoRowArgs pr{log_std, reinterpret_cast<const float4*>(action), old_lp, adv, ret, part, *cfg};
    const dim3 grid((M + 31) / 32);
    const int which = (chain_matches<NetNav>(*d) && in1 && bwd_chain_matches<NetNav, true, true, false>(*bd)) ? 2
                    : (chain_matches<NetHover>(*d) && bwd_chain_matches<NetHover, true, true, false>(*bd)) ? 1 : 0;
    if (!which) {
        for (int i = 0; i < chain_plugin_count(); ++i)      // shapes compiled on first use (vf_mlp_chain_gen.hpp)
            if (chain_plugin(i)->ppo_update)
                if (int rc = chain_plugin(i)->ppo_update(&g, &gb, &pr, M, st)) return plugin_rc(rc, "vf_ppo_update");
        return 0;
    }
    // two waves per row tile, each walking half of the network (vf_mlp_chain_split.hip); 0: switched off -> the one-wave kernel
    const int sp = ppo_update_split_try(g, gb, &pr, which, M, st);
    if (sp) return sp;
    if (which == 2) hipLaunchKernelGGL(k_ppo_update_chain<NetNav>, grid, dim3(64), 0, st, g, gb, pr);
    else hipLaunchKernelGGL(k_ppo_update_chain<NetHover>, grid, dim3(64), 0, st, g, gb, pr);
    VF_HIP(hipGetLastError());
    return 1;
}

// 1: launched, 0: the layer table is not one of the instantiated network classes, < 0: error
int mlp_forward_chain_try(const vf_mlp_desc* d, const float* params, const float* packed, const float* in0, const float* in1,
                          float* out0, float* out1, int M, hipStream_t st, const ReparamFwd* rpp, const float* in2, int M_choice)
{
    static const bool off = [] { const char* e = getenv("VISFLY_AMD_MLP_CHAIN"); return e && atoi(e) == 0; }();
    const ReparamFwd rp = rpp ? *rpp : ReparamFwd{};
    if (off || (!out0 && !rp.action) || (reinterpret_cast<uintptr_t>(out0) & 15) || (reinterpret_cast<uintptr_t>(rp.action) & 15)) return 0;
    for (int i = 0; i < d->n_layers; ++i)
        if (d->layer[i].save && !rows_fit_u32(M, d->layer[i].save_ld)) return 0;
    if (!out1) {      // no value requested: the value trunk is skipped
        if (chain_matches<NetNavPi>(*d) && in1) return chain_launch<NetNavPi>(*d, params, packed, in0, in1, out0, out1, M, st, rp, nullptr, M_choice);
        if (chain_matches<NetHoverPi>(*d)) return chain_launch<NetHoverPi>(*d, params, packed, in0, nullptr, out0, out1, M, st, rp, nullptr, M_choice);
        // (a generated class: the plugin's policy-only class below -- r06: until then this branch ended here, the caller re-ran the full class)
    } else {
        if (chain_matches<NetNav>(*d) && in1) return chain_launch<NetNav>(*d, params, packed, in0, in1, out0, out1, M, st, rp, nullptr, M_choice);
        if (chain_matches<NetHover>(*d)) return chain_launch<NetHover>(*d, params, packed, in0, nullptr, out0, out1, M, st, rp, nullptr, M_choice);
    }
    if (!in2)
        for (int i = 0; i < chain_plugin_count(); ++i)      // shapes compiled on first use (vf_mlp_chain_gen.hpp)
            if (chain_plugin(i)->forward)
                if (int rc = chain_plugin(i)->forward(d, params, packed, in0, in1, out0, out1, M, st, rpp, M_choice)) return plugin_rc(rc, "vf_mlp_forward");
    if (!rpp && out1) return mlp_forward_chain_try_sac(d, params, packed, in0, in1, in2, out0, out1, M, st, M_choice);    // SAC-style Actor / twin critic (vf_mlp_chain_sac.hip)
    return 0;
}

// the reverse chain with observation gradient a BPTT sweep runs per step: 0 none, 1 NetHover, 2 NetNav (policy trunk only), 3 NetSacHover,
// 4 NetSacNav (td_policies.Actor: both trunks);
// + 16 when M rows per pass run on the 16-rows-per-wave chain
int bwd_chain_policy_class(const vf_mlp_bwd_desc* d, int M)
{
    static const bool off = [] { const char* e = getenv("VISFLY_AMD_MLP_CHAIN"); return e && atoi(e) == 0; }();
    if (off) return 0;
    if (bwd_chain_matches<NetHover, true, false, true>(*d)) return 1 + (bwd16_ok<NetHover, true, false, true>(*d, M) ? 16 : 0);
    if (bwd_chain_matches<NetNav, true, false, true>(*d)) return 2 + (bwd16_ok<NetNav, true, false, true>(*d, M) ? 16 : 0);
    if (bwd_chain_matches<NetSacHover, true, true, true>(*d)) return 3 + (bwd16_ok<NetSacHover, true, true, true>(*d, M) ? 16 : 0);
    if (bwd_chain_matches<NetSacNav, true, true, true>(*d)) return 4 + (bwd16_ok<NetSacNav, true, true, true>(*d, M) ? 16 : 0);
    return 0;
}

// which 16-rows-per-wave policy-only class does this layer table belong to (vf_bptt_rollout.hip)?  0: none, 1: NetHoverPi
// (one observation), 2: NetNavPi (state + target), 3 / 4: NetSacHover / NetSacNav (both trunks, two 4-wide heads).  The conditions of
// chain16_ok apart from the row count.
int chain16_policy_class(const vf_mlp_desc* d, const float* params)
{
    static const bool off = [] { const char* e = getenv("VISFLY_AMD_MLP_CHAIN"); return e && atoi(e) == 0; }();
    if (off) return 0;
    if (chain_matches<NetHoverPi>(*d) && chain16_ok<NetHoverPi>(*d, params, 1)) return 1;
    if (chain_matches<NetNavPi>(*d) && chain16_ok<NetNavPi>(*d, params, 1)) return 2;
    if (chain_matches<NetSacHover>(*d) && chain16_ok<NetSacHover>(*d, params, 1)) return 3;     // td_policies.Actor: mu / log_std heads
    if (chain_matches<NetSacNav>(*d) && chain16_ok<NetSacNav>(*d, params, 1)) return 4;
    return 0;
}

// the actor-critic class of a layer table run on M rows (vf_ppo_rollout.hip): 0 none, 1 NetHover, 2 NetNav; + 16 when
// vf_mlp_forward would run those M rows on the 16-rows-per-wave chain
int chain_full_class(const vf_mlp_desc* d, const float* params, int M)
{
    static const bool off = [] { const char* e = getenv("VISFLY_AMD_MLP_CHAIN"); return e && atoi(e) == 0; }();
    if (off) return 0;
    if (chain_matches<NetHover>(*d)) return 1 + (chain16_ok<NetHover>(*d, params, M) ? 16 : 0);
    if (chain_matches<NetNav>(*d)) return 2 + (chain16_ok<NetNav>(*d, params, M) ? 16 : 0);
    return 0;
}

}  // namespace vf

#ifdef VF_CHAIN_TRACE
extern "C" int vf_debug_chain_trace(long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(vf::vf_chain_trace), sizeof(long long) * 64, 0, hipMemcpyDeviceToHost);
}
#endif
