// vf_bptt_reverse_kernel.hpp -- k_bptt_reverse (the reverse half of a BPTT horizon as one persistent launch; scheme at the head of
// vf_bptt_reverse.hip) + its instance table, shared by vf_bptt_reverse.hip (MlpPolicy classes) and vf_bptt_reverse_sac.hip
// (td_policies.Actor), two translation units so that their instances compile side by side.
#pragma once
#include "vf_env_bwd_body.hpp"
#include "vf_mlp_chain_bwd.hpp"

#pragma clang fp contract(off)

namespace vf {

#ifdef VF_PPO_TRACE
__device__ long long vf_rev_trace[8];
#endif

struct RevArgs {
    int H, N, G, g_drag, g_race;
    const float* tape;             // [H] rows of tape_stride floats: the slab before step t
    long long tape_stride;
    const float4* actions;         // [H][N]: the action step t was given
    const unsigned char* done;     // [H][N]
    const float* d_reward;         // [H][N]
    float* adj;                    // adjoint of the persistent state (slab layout), in / out
    float4* d_action;              // [H][N] scratch: dLoss / d action_t, read by the head reverse of step t
    const float* g_obs;            // [H][N][13]: row t N + i = dLoss / d (observation of slot t), written by the reverse chain
    const float4* ck;              // the forward launch's sub-step tape [H][S + 3][waves of 16 agents][64] float4 (CKPT instances), else null
};

template <class P, int ROWS, int KIND, int ACT, int INTEG, bool CTRL_DELAY, bool CKPT>
__global__ __launch_bounds__(64) void k_bptt_reverse(const vf_dyn_cfg* __restrict__ cp, const vf_env_cfg* __restrict__ ep, const BwdArgsChain gb,
                                                     const RevArgs r)
{
    prefetch_kernarg<sizeof(BwdArgsChain) + sizeof(RevArgs) + 16>();
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [S * kSave][64]
    const int lane = threadIdx.x, m = lane & (ROWS - 1);
    const int i = min((int)blockIdx.x * ROWS + m, r.N - 1);          // lanes past the last agent replicate it as well
#ifdef VF_PPO_TRACE
    long long tr[2] = {0, 0}, tc = __builtin_readcyclecounter();
#define VF_RT(k) do { const long long n_ = __builtin_readcyclecounter(); tr[k] += n_ - tc; tc = n_; } while (0)
#else
#define VF_RT(k) do { } while (0)
#endif
    // CKPT: this wave's record of step t travels global -> LDS by LDS-DMA (no registers; one 1 KiB row per instruction) a whole
    // step AHEAD of its use, into the half of the LDS area the current step does not read: issued at the head of step t + 1, it has
    // ~28 us to arrive from HBM (the tape was written a forward sweep ago) and nothing ever waits for it -- fetched by the step
    // that needs it, it cost as much latency as the replay it replaces (r04: reverse half 28.4 -> 28.6 us per step)
    const int rows_ck = cp->interval_steps + 3;      // S sub-step heads, the pre-clamp end state, the step's inputs, its drag granules
    float4* lds4 = reinterpret_cast<float4*>(lds);
    auto fetch_record = [&](int t) {
        const float4* src = r.ck + ((size_t)t * rows_ck * gridDim.x + blockIdx.x) * 64 + lane;
        float4* dst = lds4 + (size_t)(t & 1) * rows_ck * 64;
        for (int j = 0; j < rows_ck; ++j)
            __builtin_amdgcn_global_load_lds(src + (size_t)j * gridDim.x * 64, (__attribute__((address_space(3))) void*)(dst + (size_t)j * 64), 16, 0, 0);
    };
    // CKPT (four lanes per agent, vf_env_bwd_quad.hpp): the adjoint of the persistent state stays in registers for the whole sweep
    // (component layout: 6 + delay_steps registers), d_action_t and dLoss / d obs_t travel between the adjoint and the chain through
    // LDS -- no slab round trip, no global hand-over, hence no fence that waits for the step's stores to be acknowledged
    const int iq = min((int)blockIdx.x * ROWS + (lane >> 2), r.N - 1);      // the quads (lanes 4 m .. 4 m + 3) hold the agent the chain keeps in lanes m + 16 kq
    float* da_lds = lds + (size_t)2 * rows_ck * 256;                         // [16 slots] float4
    float* obs_lds = da_lds + 64;                                            // [16 slots][16] floats
    float std_reg[4] = {0.0f, 0.0f, 0.0f, 0.0f};      // exp(log_std) of the state-independent head, once per launch (see k_bptt_rollout)
    if constexpr (!P::sac_head) {
#pragma unroll
        for (int k = 0; k < 4; ++k) std_reg[k] = expf(gb.rp_log_std[k]);
    }
    QuadCarry cy;
    QuadLane ql;
    if constexpr (CKPT) {
        ql = quad_lane(*cp, lane);
        fetch_record(r.H - 1);
        const int k = lane & 3;
        const float4 g0 = *granule(r.adj, r.G, iq, VF_G_POS), g1 = *granule(r.adj, r.G, iq, VF_G_QUAT), g2 = *granule(r.adj, r.G, iq, VF_G_VEL);
        const float4 g3 = *granule(r.adj, r.G, iq, VF_G_OMG), g4 = *granule(r.adj, r.G, iq, VF_G_MOT), g6 = *granule(r.adj, r.G, iq, VF_G_AACC);
        cy.lp = q_sel4(k, 0.0f, g0.y, g0.z, g0.w);
        cy.lq = q_sel4(k, g1.x, g1.y, g1.z, g1.w);
        cy.lv = q_sel4(k, 0.0f, g2.y, g2.z, g2.w);
        cy.lw = q_sel4(k, 0.0f, g3.y, g3.z, g3.w);
        cy.lwm = q_sel4(k, g4.x, g4.y, g4.z, g4.w);
        cy.laa = q_sel4(k, 0.0f, g6.y, g6.z, g6.w);
#pragma unroll
        for (int q = 0; q < kRingRegs; ++q) {
            cy.ring[q] = 0.0f;
            if (q < cp->delay_steps) {
                const float4 gr = *granule(r.adj, r.G, iq, VF_G_RING + q);
                cy.ring[q] = q_sel4(k, gr.x, gr.y, gr.z, gr.w);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (int t = r.H - 1; t >= 0; --t) {
        const BwdArgs g{r.N, r.G, r.g_drag, r.g_race, r.tape + (size_t)t * r.tape_stride, r.actions + (size_t)t * r.N,
                        t + 1 < r.H ? r.g_obs + (size_t)(t + 1) * r.N * obs_width(KIND) : nullptr, r.d_reward + (size_t)t * r.N,
                        r.done + (size_t)t * r.N, r.adj, r.d_action + (size_t)t * r.N};
        const int row = t * r.N + i;
        // the masks of this step's reverse chain (saved activations of slot t: written a forward sweep ago, HBM by now) are loaded
        // HERE, ahead of the adjoint, which covers their latency; issued inside the chain, whose ops are over long before their
        // own loads are back, they cost 5 of the step's 31 us
        BwdState16<P> st16;
        if constexpr (ROWS == 16) bwd16_mask_preload<P, 0>(gb, st16, row, lane >> 4);
        if constexpr (CKPT) {
            if (t > 0) fetch_record(t - 1);
            env_step_bwd_agent<KIND, ACT, INTEG, CTRL_DELAY, 64, true, true>(*cp, *ep, g, iq, true, lds + lane, lds4 + (size_t)(t & 1) * rows_ck * 64, &cy,
                                                                             obs_lds, da_lds, &ql);
            // the record of step t - 1 and this step's masks were issued at the head of the step and the adjoint issued no other
            // vector-memory operation: they are back by now (on gfx9 vmcnt also counts STORES: waiting here, not behind the chain,
            // keeps the chain's 40-odd dZ stores out of the wait).  LDS operations of a wave execute in order.
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            st16.pda = reinterpret_cast<const float4*>(da_lds)[lane & 15];
        } else {
            env_step_bwd_agent<KIND, ACT, INTEG, CTRL_DELAY, 64>(*cp, *ep, g, i, true, lds + lane);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // d_action_t: written above, read by the head reverse below
            if constexpr (ROWS == 16) st16.pda = r.d_action[(size_t)t * r.N + i];
        }
        VF_RT(0);
        // (an opaque copy of the lane id per iteration: the chain's loop-invariant per-item load offsets stay just-in-time instead
        // of being hoisted out of the t loop into ~100 live registers -- see k_ppo_rollout)
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        long zero_t = 0;                                             // (likewise for the per-item weight base addresses)
        asm volatile("" : "+s"(zero_t));
        BwdArgsChain gbt = gb;
        gbt.packed = gb.packed + zero_t;
        gbt.rp_std_valid = 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) gbt.rp_std[k] = std_reg[k];
        if constexpr (ROWS == 16) {
            const int gq = lane_t >> 4;
            bwd16_prologue<P, 0>(gbt, st16, lane_t);
            bwd16_head_prologue<P, 0, true>(gbt, st16, row, gq, true);
            bwd16_items<P, 0, true>(gbt, st16, lane_t, row, row, true);
            bwd16_tail_store<P>(gbt, st16, row, gq, true);
            if constexpr (CKPT) {      // dLoss / d obs_t for the adjoint of step t - 1: features 4 gq .. 4 gq + 3 of row lane & 15
                const f32x4& v = bwd16_obs_tile<P>(st16);
                *reinterpret_cast<float4*>(obs_lds + (lane_t & 15) * 16 + 4 * gq) = make_float4(v[0], v[1], v[2], v[3]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else {
            bwd_rows<P, ROWS>(gbt, lane_t, row, row, true);
        }
        if constexpr (!CKPT) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");       // dLoss / d obs_t: read by the adjoint of step t - 1
        VF_RT(1);
    }
    if constexpr (CKPT) {      // the adjoint of the state before the first step of the horizon, back into the slab
        *granule(r.adj, r.G, iq, VF_G_POS) = make_float4(0.f, qb<1>(cy.lp), qb<2>(cy.lp), qb<3>(cy.lp));
        *granule(r.adj, r.G, iq, VF_G_QUAT) = make_float4(qb<0>(cy.lq), qb<1>(cy.lq), qb<2>(cy.lq), qb<3>(cy.lq));
        *granule(r.adj, r.G, iq, VF_G_VEL) = make_float4(0.f, qb<1>(cy.lv), qb<2>(cy.lv), qb<3>(cy.lv));
        *granule(r.adj, r.G, iq, VF_G_OMG) = make_float4(0.f, qb<1>(cy.lw), qb<2>(cy.lw), qb<3>(cy.lw));
        *granule(r.adj, r.G, iq, VF_G_MOT) = make_float4(qb<0>(cy.lwm), qb<1>(cy.lwm), qb<2>(cy.lwm), qb<3>(cy.lwm));
        *granule(r.adj, r.G, iq, VF_G_THR) = make_float4(0.f, 0.f, 0.f, 0.f);
        *granule(r.adj, r.G, iq, VF_G_AACC) = make_float4(0.f, qb<1>(cy.laa), qb<2>(cy.laa), qb<3>(cy.laa));
        *granule(r.adj, r.G, iq, VF_G_ACC) = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < kRingRegs; ++q)
            if (q < cp->delay_steps)
                *granule(r.adj, r.G, iq, VF_G_RING + q) = make_float4(qb<0>(cy.ring[q]), qb<1>(cy.ring[q]), qb<2>(cy.ring[q]), qb<3>(cy.ring[q]));
    }
#ifdef VF_PPO_TRACE
    if (blockIdx.x == 7 && lane == 0) { vf_rev_trace[0] = tr[0]; vf_rev_trace[1] = tr[1]; }
#endif
}

}  // namespace vf

namespace vf {

using RevKernel = void (*)(const vf_dyn_cfg*, const vf_env_cfg*, const vf::BwdArgsChain, const vf::RevArgs);

// DELAY: the motor-lag form of the interval or the direct one (pick_roll, vf_bptt_rollout_kernel.hpp); the direct instances live in
// vf_bptt_reverse_nodelay.hip
template <class Net, int ROWS, int KIND, bool CKPT, bool DELAY>
static RevKernel pick_rev2(const vf_dyn_cfg& c)
{
    using P = vf::BwdProg<Net, true, Net::HV == 4, true>;      // HV == 4: td_policies.Actor, both trunks
    if ((c.ctrl_delay != 0) != DELAY) return nullptr;
    if (c.integrator == VF_INT_RK4) {
        if (c.action_type == VF_ACT_THRUST) return vf::k_bptt_reverse<P, ROWS, KIND, VF_ACT_THRUST, VF_INT_RK4, DELAY, CKPT>;
        if (c.action_type == VF_ACT_BODYRATE) return vf::k_bptt_reverse<P, ROWS, KIND, VF_ACT_BODYRATE, VF_INT_RK4, DELAY, CKPT>;
        return nullptr;
    }
    if (c.action_type == VF_ACT_THRUST) return vf::k_bptt_reverse<P, ROWS, KIND, VF_ACT_THRUST, VF_INT_EULER, DELAY, CKPT>;
    if (c.action_type == VF_ACT_BODYRATE) return vf::k_bptt_reverse<P, ROWS, KIND, VF_ACT_BODYRATE, VF_INT_EULER, DELAY, CKPT>;
    return nullptr;
}

// ckpt: the forward launch wrote the sub-step tape -> the instances that read it instead of replaying the interval
template <class Net, int ROWS, int KIND, bool DELAY = true>
static RevKernel pick_rev(const vf_dyn_cfg& c, bool ckpt)
{
    if constexpr (ROWS == 16) {          // the tape's records are the forward launch's waves: 16 agents each
        if (ckpt) return pick_rev2<Net, ROWS, KIND, true, DELAY>(c);
    }
    return pick_rev2<Net, ROWS, KIND, false, DELAY>(c);
}

// vf_bptt_reverse_nav2.hip: the one-observation classes (net 1, 3) over the Navigation env kind (NavigationEnv2); both forms of the interval
RevKernel pick_rev_nav2(int net, bool r16, const vf_dyn_cfg& c, bool ckpt);
// vf_bptt_reverse_race2.hip: the one-observation classes (net 1, 3) over RacingEnv2's 16-column rows (kernel-side kind VF_ENV_RACING2)
RevKernel pick_rev_race2(int net, bool r16, const vf_dyn_cfg& c, bool ckpt);
// vf_bptt_reverse_nodelay.hip: every class with ctrl_delay = false (net: bwd_chain_policy_class's 1 .. 4; r16: 16 rows per wave)
RevKernel pick_rev_nodelay(int net, bool r16, int kind, const vf_dyn_cfg& c, bool ckpt);

// vf_bptt_reverse_sac.hip: net = 3 NetSacHover (Hover / Racing env), 4 NetSacNav (Navigation env); 16 rows per wave, and (r05) 32 for
// N > 16 384 agents per launch
RevKernel pick_rev_sac(int net, int kind, const vf_dyn_cfg& c, bool ckpt);
RevKernel pick_rev_sac32(int net, int kind, const vf_dyn_cfg& c);

}  // namespace vf


namespace vf {

// layout stamp of what a BPTT plugin's reverse sweep is handed (vf_chain_plugin.hpp: ChainPlugin::bptt_rev_abi)
constexpr unsigned kBpttRevPluginAbi = 0x42560001u ^ (unsigned)(sizeof(BwdArgsChain) * 31u + sizeof(RevArgs) * 17u + sizeof(vf_dyn_cfg) * 7u +
                                                               sizeof(vf_env_cfg) * 5u);

}  // namespace vf

#if defined(VF_CHAIN_PLUGIN) && (VF_CHAIN_PLUGIN_PART == 6 || VF_CHAIN_PLUGIN_PART == 7)
#include "vf_mlp_chain_gen.hpp"
#include "vf_chain_plugin.hpp"
extern "C" int vf_plugin_bptt_reverse(const vf_mlp_bwd_desc*, int, const vf_dyn_cfg*, const vf_dyn_cfg*, const vf_env_cfg*, const vf::BwdArgsChain*,
                                      const void*, int, size_t, hipStream_t);
#endif
#if defined(VF_CHAIN_PLUGIN) && VF_CHAIN_PLUGIN_PART == 6
namespace vf {

// the reverse half for the same class: 16 agents per wave, the sub-step tape of the forward launch (CKPT); P = the reverse chain a BPTT
// sweep runs per step (observation gradient; both trunks for the SAC-style Actor, the policy trunk for an actor-critic)
template <class P, int KIND, int ACT, int INTEG, bool DELAY>
int plugin_bptt_reverse(const vf_mlp_bwd_desc* d, int env_kind, const vf_dyn_cfg* c, const vf_dyn_cfg* d_dyn, const vf_env_cfg* d_env,
                        const BwdArgsChain* gb, const void* rev_args, int N, size_t lds, hipStream_t st)
{
    if (env_kind != KIND || c->action_type != ACT || c->integrator != INTEG || (c->ctrl_delay != 0) != DELAY) return 0;
    if (!bwd_chain_matches_gen<P>(*d, true) || !bwd16_ok_gen<P>(*d, N)) return 0;
    if (P::sac_head ? !gb->rp_ls_rows : (!gb->rp_log_std || !gb->rp_g_log_std)) return 0;
    hipLaunchKernelGGL((k_bptt_reverse<P, 16, KIND, ACT, INTEG, DELAY, true>), dim3((N + 15) / 16), dim3(64), lds, st, d_dyn, d_env, *gb,
                       *static_cast<const RevArgs*>(rev_args));
    VF_HIP(hipGetLastError());
    return 1;
}

}  // namespace vf

#define VF_CHAIN_PLUGIN_BPTT_DEFINE(Net, NetPi, KIND, ACT, INTEG, DELAY, NAME)                                                                  \
    extern "C" int vf_plugin_bptt_reverse(const vf_mlp_bwd_desc* d, int env_kind, const vf_dyn_cfg* c, const vf_dyn_cfg* d_dyn,                 \
                                          const vf_env_cfg* d_env, const vf::BwdArgsChain* gb, const void* ra, int N, size_t lds,               \
                                          hipStream_t st)                                                                                       \
    {                                                                                                                                           \
        using P = typename Net::template Bwd<true, Net::HV == 4, true>;                                                                         \
        return vf::plugin_bptt_reverse<P, KIND, ACT, INTEG, DELAY>(d, env_kind, c, d_dyn, d_env, gb, ra, N, lds, st);                           \
    }
#endif
#if defined(VF_CHAIN_PLUGIN) && VF_CHAIN_PLUGIN_PART == 7
#include "vf_bptt_rollout_kernel.hpp"
#define VF_CHAIN_PLUGIN_BPTT_DEFINE(Net, NetPi, KIND, ACT, INTEG, DELAY, NAME)                                                                  \
    extern "C" const vf::ChainPlugin* vf_chain_plugin()                                                                                         \
    {                                                                                                                                           \
        static const vf::ChainPlugin p{vf::kChainPluginAbi, NAME, nullptr, nullptr, nullptr, nullptr, 0u, nullptr,                              \
                                       vf::kBpttRollPluginAbi, vf::kBpttRevPluginAbi, vf_plugin_bptt_rollout, vf_plugin_bptt_reverse};          \
        return &p;                                                                                                                              \
    }
#endif
