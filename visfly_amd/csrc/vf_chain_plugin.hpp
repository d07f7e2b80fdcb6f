// vf_chain_plugin.hpp -- the interface between libvisfly_amd.so and a chain plugin: a shared object, compiled on first use for one network
// shape (visfly_amd/_jit.py), that holds the register-chained kernels of that shape (vf_mlp_chain_gen.hpp) and the three host functions
// below.  libvisfly_amd.so loads it with vf_chain_plugin_load (include/visfly_amd.h) and asks every loaded plugin after its own classes
// (mlp_forward_chain_try, mlp_backward_chain_try, ppo_update_chain_try).  Return values as those functions': 1 launched (or, for a
// query, "would launch"), 0 not this plugin's shape, < 0 error (-1000 - hipError_t).
#pragma once
#include "vf_mlp_chain_kernels.hpp"
#ifdef VF_CHAIN_PLUGIN
#include "vf_mlp_chain_gen.hpp"
#endif

namespace vf {

// layout stamp: both sides are compiled from the same headers; a plugin built against other struct layouts is refused
constexpr unsigned kChainPluginAbi = 0x56460003u ^ (unsigned)(sizeof(vf_mlp_desc) * 31u + sizeof(vf_mlp_bwd_desc) * 17u + sizeof(ChainArgs) * 7u +
                                                            sizeof(BwdArgsChain) * 5u + sizeof(PpoRowArgs) * 3u + sizeof(ReparamFwd) + sizeof(ReparamBwd));

struct ChainPlugin {
    unsigned abi;
    const char* name;      // the shape, for messages
    // out1 == null: the policy-only class (no value trunk).  M_choice > 0: the rows-per-wave choice is made for M_choice rows
    // (vf_mlp_forward_steps).  The classes a BPTT / SHAC horizon steps -- the SAC-style Actor, the policy-only class -- run 16 rows per wave
    // where the built-in classes do (chain16_ok), so that their persistent launches (16 agents per wave) equal the per-step path to the bit
    int (*forward)(const vf_mlp_desc* d, const float* params, const float* packed, const float* in0, const float* in1, float* out0, float* out1,
                   int M, hipStream_t st, const ReparamFwd* rp, int M_choice);
    // packed == null: capability query
    int (*backward)(const vf_mlp_bwd_desc* d, const float* packed, int M, hipStream_t st, const ReparamBwd* rp);
    int (*ppo_update)(const ChainArgs* g, const BwdArgsChain* gb, const PpoRowArgs* pr, int M, hipStream_t st);
    // the twin critic's classes (heads (1, 1), pass-through action tile): SHAC's fused critic step (k_twin_q_update_chain); null: not a critic
    int (*twin_q_update)(const ChainArgs* g, const BwdArgsChain* gb, const float* target, double* part, float scale, int M, hipStream_t st);
    // a ROLL-OUT plugin (one more shared object per shape AND env kind / action type / integrator / ctrl_delay: the persistent
    // launch of vf_ppo_rollout.hip is a template over all of them) sets only this one; env_args / roll_args: vf::EnvArgs /
    // vf::PpoRollArgs (vf_ppo_rollout_kernel.hpp; rollout_abi stamps their layout), c: the host copy of the dynamics constants
    unsigned rollout_abi;
    int (*ppo_rollout)(const vf_mlp_desc* d, int env_kind, const vf_dyn_cfg* c, int has_target, const vf_dyn_cfg* d_dyn, const vf_env_cfg* d_env,
                       const void* env_args, const ChainArgs* gc, const void* roll_args, int N, hipStream_t st);
    // a BPTT plugin (r06; per shape AND env kind / action type / integrator / ctrl_delay like the roll-out plugin): the two persistent
    // launches of a BPTT / SHAC horizon for a generated actor class, 16 agents per wave -- k_bptt_rollout (vf_bptt_rollout_kernel.hpp:
    // env_args / roll_args = vf::EnvArgs / vf::RollArgs, stamped by bptt_roll_abi) and k_bptt_reverse with the sub-step tape
    // (vf_bptt_reverse_kernel.hpp: rev_args = vf::RevArgs, bptt_rev_abi; lds = its dynamic LDS bytes).  env_kind: the KERNEL-side kind
    // (VF_ENV_RACING2 for RacingEnv2's 16-column rows)
    unsigned bptt_roll_abi, bptt_rev_abi;
    int (*bptt_rollout)(const vf_mlp_desc* d, const float* params, int env_kind, const vf_dyn_cfg* c, int has_target, const vf_dyn_cfg* d_dyn,
                        const vf_env_cfg* d_env, const void* env_args, const ChainArgs* gc, const void* roll_args, int N, hipStream_t st);
    int (*bptt_reverse)(const vf_mlp_bwd_desc* d, int env_kind, const vf_dyn_cfg* c, const vf_dyn_cfg* d_dyn, const vf_env_cfg* d_env,
                        const BwdArgsChain* gb, const void* rev_args, int N, size_t lds, hipStream_t st);
};

// the registry (vf_chain_plugin.hip)
int chain_plugin_count();
const ChainPlugin* chain_plugin(int i);
void chain_plugin_count_launch();      // vf_chain_plugin_launches(): how tests see that a plugin, not the block-tile kernels, served a call

}  // namespace vf

#ifdef VF_CHAIN_PLUGIN
// ---- the plugin side: VF_CHAIN_PLUGIN_PART selects what this translation unit compiles (the parts compile in parallel) ----
//   1: forward kernels, 2: reverse chains, 3: fused PPO step, 0: the table
namespace vf {

template <class Net, class NetPi>
int plugin_forward(const vf_mlp_desc* d, const float* params, const float* packed, const float* in0, const float* in1, float* out0, float* out1, int M,
                   hipStream_t st, const ReparamFwd* rpp, int M_choice)
{
    if (Net::NB + Net::PASS == 2 && !in1) return 0;
    const ReparamFwd rp = rpp ? *rpp : ReparamFwd{};
    if (Net::PASS && (rpp || !out0 || !out1)) return 0;      // the twin critic: Q1 / Q2 (M,), no action head
    ChainArgs g{*d, params, packed, ChainIo{{in0, in1, nullptr}, out0, out1}, M, rp.log_std, reinterpret_cast<const float4*>(rp.eps),
                reinterpret_cast<float4*>(rp.action), {rp.obs_copy[0], rp.obs_copy[1]}};
    if (!out1) {
        if constexpr (Net::HV != 1 || Net::HM != 4) {
            return 0;             // (the SAC-style Actor always runs both trunks)
        } else {
            if (!chain_matches_gen<NetPi>(*d)) return 0;
            // with the action head (vf_mlp_forward_act: what a BPTT horizon steps) 16 rows per wave for small row counts, like the
            // persistent launch that replaces the loop (k_bptt_rollout); the plain policy-only forward keeps the full class's 32 rows,
            // whose mean it reproduces bit for bit
            if (rp.action && chain16_ok<NetPi>(*d, params, M_choice > 0 ? M_choice : M))
                hipLaunchKernelGGL(k_mlp_forward_chain16<NetPi>, dim3((M + 15) / 16), dim3(64), 0, st, g);
            else
                hipLaunchKernelGGL(k_mlp_forward_chain<NetPi>, dim3((M + 31) / 32), dim3(64), 0, st, g);
        }
    } else {
        if (!chain_matches_gen<Net>(*d)) return 0;
        if constexpr (Net::HV == 4) {      // the SAC-style Actor: 16 rows per wave for small row counts (chain_launch's rule)
            if (chain16_ok<Net>(*d, params, M_choice > 0 ? M_choice : M)) {
                hipLaunchKernelGGL(k_mlp_forward_chain16<Net>, dim3((M + 15) / 16), dim3(64), 0, st, g);
                VF_HIP(hipGetLastError());
                return 1;
            }
        }
        hipLaunchKernelGGL(k_mlp_forward_chain<Net>, dim3((M + 31) / 32), dim3(64), 0, st, g);
    }
    VF_HIP(hipGetLastError());
    return 1;
}

template <class Net>
int plugin_backward(const vf_mlp_bwd_desc* d, const float* packed, int M, hipStream_t st, const ReparamBwd* rpp)
{
    using PU = typename Net::template Bwd<true, true, false>;      // PPO update / SHAC actor: both trunks, no observation gradient
    // with observation gradient: the policy trunk alone (first-order optimisation of an actor-critic's policy) or, for the SAC-style
    // Actor, both trunks (mu and log_std heads both carry gradient)
    // (the twin critic has no such variant: PG = PU)
    using PG = std::conditional_t<Net::PASS != 0, PU, typename Net::template Bwd<true, Net::HV == 4, true>>;
    const ReparamBwd rp = rpp ? *rpp : ReparamBwd{};
    BwdArgsChain g{*d, packed, M, reinterpret_cast<const float4*>(rp.d_action), reinterpret_cast<const float4*>(rp.action), rp.log_std,
                   reinterpret_cast<const float4*>(rp.eps), reinterpret_cast<float4*>(rp.g_log_std)};
    const dim3 grid((M + 31) / 32);
    if (bwd_chain_matches_gen<PU>(*d, false)) {
        if (packed) hipLaunchKernelGGL(k_mlp_backward_chain<PU>, grid, dim3(64), 0, st, g);
    } else if (!Net::PASS && bwd_chain_matches_gen<PG>(*d, true)) {
        if constexpr (!Net::PASS) {        // what a BPTT sweep runs per step: 16 rows per wave for small row counts (bwd_chain_launch's rule)
            if (packed && bwd16_ok_gen<PG>(*d, M)) {
                hipLaunchKernelGGL((k_mlp_backward_chain<PG, 16>), dim3((M + 15) / 16), dim3(64), 0, st, g);
                VF_HIP(hipGetLastError());
                return 1;
            }
        }
        if (packed) hipLaunchKernelGGL(k_mlp_backward_chain<PG>, grid, dim3(64), 0, st, g);
    } else {
        return 0;
    }
    VF_HIP(hipGetLastError());
    return 1;
}

template <class Net0>
int plugin_ppo_update(const ChainArgs* g, const BwdArgsChain* gb, const PpoRowArgs* pr, int M, hipStream_t st)
{
    // more forward tiles than stay live beside the reverse chain: the variant that keeps the ReLU masks as bits (vf_mlp_chain_gen.hpp)
    // (a Tanh / ELU / LeakyReLU network needs the values: above the live-tile limit it has no fused step -- forward, loss and reverse chain
    // then run as the chain launches of their own)
    constexpr bool packable = Net0::all_relu;
    using Net = std::conditional_t<(Net0::n_tiles > kGenLiveTiles && packable), ChainNetG<typename Net0::Spec, true>, Net0>;
    if constexpr (Net::HV != 1 || Net::HM != 4 || (Net0::n_tiles > kGenLiveTiles && !packable)) {
        return 0;                 // (no PPO step on the SAC-style Actor / the twin critic: the kernel is not instantiated)
    } else {
        using PU = typename Net::template Bwd<true, true, false>;
        if (Net::NB == 2 && !g->io.in[1]) return 0;
        if (!chain_matches_gen<Net>(g->d) || !bwd_chain_matches_gen<PU>(gb->d, false)) return 0;
        hipLaunchKernelGGL(k_ppo_update_chain<Net>, dim3((M + 31) / 32), dim3(64), 0, st, *g, *gb, *pr);
        VF_HIP(hipGetLastError());
        return 1;
    }
}

// SHAC's fused critic step on a generated twin-critic class (vf_mlp_chain_sac.hip: twin_q_update_chain_try checked the save pointers / row counts)
template <class Net0>
int plugin_twin_q_update(const ChainArgs* g, const BwdArgsChain* gb, const float* target, double* part, float scale, int M, hipStream_t st)
{
    constexpr bool packable = Net0::all_relu;
    using Net = std::conditional_t<(Net0::n_tiles > kGenLiveTiles && packable), ChainNetG<typename Net0::Spec, true>, Net0>;
    if constexpr (!Net::PASS || (Net0::n_tiles > kGenLiveTiles && !packable)) {
        return 0;
    } else {
        using PU = typename Net::template Bwd<true, true, false>;
        if (!g->io.in[1] || !chain_matches_gen<Net>(g->d) || !bwd_chain_matches_gen<PU>(gb->d, false)) return 0;
        hipLaunchKernelGGL(k_twin_q_update_chain<Net>, dim3((M + 31) / 32), dim3(64), 0, st, *g, *gb, target, part, scale);
        VF_HIP(hipGetLastError());
        return 1;
    }
}

}  // namespace vf

extern "C" {
int vf_plugin_twin_q_update(const vf::ChainArgs*, const vf::BwdArgsChain*, const float*, double*, float, int, hipStream_t);
int vf_plugin_forward(const vf_mlp_desc*, const float*, const float*, const float*, const float*, float*, float*, int, hipStream_t, const vf::ReparamFwd*, int);
int vf_plugin_backward(const vf_mlp_bwd_desc*, const float*, int, hipStream_t, const vf::ReparamBwd*);
int vf_plugin_ppo_update(const vf::ChainArgs*, const vf::BwdArgsChain*, const vf::PpoRowArgs*, int, hipStream_t);
const vf::ChainPlugin* vf_chain_plugin();
}

// VF_CHAIN_PLUGIN_DEFINE(Net, NetPi, "name"): the part of the plugin this translation unit holds
#if VF_CHAIN_PLUGIN_PART == 1
#define VF_CHAIN_PLUGIN_DEFINE(Net, NetPi, NAME)                                                                                             \
    extern "C" int vf_plugin_forward(const vf_mlp_desc* d, const float* params, const float* packed, const float* in0, const float* in1,    \
                                     float* out0, float* out1, int M, hipStream_t st, const vf::ReparamFwd* rp, int M_choice)                \
    { return vf::plugin_forward<Net, NetPi>(d, params, packed, in0, in1, out0, out1, M, st, rp, M_choice); }
#elif VF_CHAIN_PLUGIN_PART == 2
#define VF_CHAIN_PLUGIN_DEFINE(Net, NetPi, NAME)                                                                                             \
    extern "C" int vf_plugin_backward(const vf_mlp_bwd_desc* d, const float* packed, int M, hipStream_t st, const vf::ReparamBwd* rp)        \
    { return vf::plugin_backward<Net>(d, packed, M, st, rp); }
#elif VF_CHAIN_PLUGIN_PART == 3
#define VF_CHAIN_PLUGIN_DEFINE(Net, NetPi, NAME)                                                                                             \
    extern "C" int vf_plugin_ppo_update(const vf::ChainArgs* g, const vf::BwdArgsChain* gb, const vf::PpoRowArgs* pr, int M, hipStream_t st) \
    { return vf::plugin_ppo_update<Net>(g, gb, pr, M, st); }                                                                                 \
    extern "C" int vf_plugin_twin_q_update(const vf::ChainArgs* g, const vf::BwdArgsChain* gb, const float* target, double* part,            \
                                           float scale, int M, hipStream_t st)                                                               \
    { return vf::plugin_twin_q_update<Net>(g, gb, target, part, scale, M, st); }
#elif VF_CHAIN_PLUGIN_PART == 4
// the roll-out plugin: one translation unit, one kernel instance (vf_ppo_rollout_kernel.hpp defines VF_CHAIN_PLUGIN_ROLLOUT_DEFINE)
#define VF_CHAIN_PLUGIN_DEFINE(Net, NetPi, NAME)
#elif VF_CHAIN_PLUGIN_PART >= 5
// the BPTT plugin: parts 5 (k_bptt_rollout), 6 (k_bptt_reverse), 7 (the table): vf_bptt_rollout_kernel.hpp / vf_bptt_reverse_kernel.hpp
// define VF_CHAIN_PLUGIN_BPTT_DEFINE
#define VF_CHAIN_PLUGIN_DEFINE(Net, NetPi, NAME)
#else
#define VF_CHAIN_PLUGIN_DEFINE(Net, NetPi, NAME)                                                                                             \
    extern "C" const vf::ChainPlugin* vf_chain_plugin()                                                                                      \
    {                                                                                                                                        \
        static const vf::ChainPlugin p{vf::kChainPluginAbi, NAME, vf_plugin_forward, vf_plugin_backward, vf_plugin_ppo_update,               \
                                       Net::PASS ? vf_plugin_twin_q_update : nullptr, 0u, nullptr, 0u, 0u, nullptr, nullptr};                \
        return &p;                                                                                                                           \
    }
#endif
#endif
