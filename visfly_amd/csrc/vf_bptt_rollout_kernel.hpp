// vf_bptt_rollout_kernel.hpp -- k_bptt_rollout (the forward half of a BPTT horizon as one persistent launch; scheme at the head of
// vf_bptt_rollout.hip) + its instance table, shared by vf_bptt_rollout.hip (state-independent log_std: the MlpPolicy classes) and
// vf_bptt_rollout_sac.hip (td_policies.Actor: state-dependent log_std), two translation units so that their instances compile side by side.
#pragma once
#include "vf_env_epilogue.hpp"
#include "vf_dyn_quad.hpp"
#include "vf_mlp_chain.hpp"

#pragma clang fp contract(off)

namespace vf {

struct RollArgs {
    int H;                     // control steps
    int N;                     // agents = policy rows per step
    float* tape;               // [H][slab floats]: row t = the slab before step t (what vf_env_step_bwd reads)
    long long tape_stride;     // floats between two tape rows
    unsigned char* tape_done;  // [H][N]
    float* d_reward;           // [H][N]  -disc_t * scale
    float* loss;               // [N] in/out
    float* disc;               // [N] in/out
    float* obs_slots;          // [H][N][13]: slot t = the observation the policy sees at step t (slot 0 filled by the caller)
    float* obs_final;          // (N,13): the observation after the last step
    float gamma, scale;
    float4* ck;                // optional sub-step tape [H][S + 3][waves][64] float4 (include/visfly_amd.h, vf_bptt_rollout) or null
    float* reward_rows;        // optional [H][N]: the reward of every step (SHAC's horizon buffer, shac.py:259-266)
    unsigned char* ep_flag_rows;   // optional [H][N]: out->ep_flags of the step, written where done (shac.py:231-232 reads episode_done there)
};

// control_interval_quad observer (component layout: lane 4 m + k holds component k of agent slot m's quantities): the agent at the
// head of every sub-step and the state after the last one before the clamps -- what the adjoint of the interval otherwise obtains by
// replaying it.  One record row = ONE store instruction of the whole wave, 1 KiB contiguous: the float4 at [k * 16 + m] of a row holds
// component k of four quantities of slot m, vectors as pure quaternions (0, x, y, z):
//     sub-step rows   (q_k, v_k, w_k, rotor speed k)        end row   (p_k, q_k, v_k, w_k)
// so that the reverse sweep, in the same layout, reads ONE float4 per lane and sub-step; k_bptt_reverse fetches a row with one LDS-DMA.
struct TapeCheckpoint {
    float4* p;                 // row 0 of this (step, wave) + this lane's position (lane & 3) * 16 + (lane >> 2), or null (no tape)
    size_t rs;                 // float4 between two rows of a record = 64 x waves
    int S, k;                  // sub-steps per interval; this lane's component / entry (lane & 3)
    // two-level selects on the bits of k (a chain `k == 0 ? .. : k == 1 ? ..` is compiled into a scratch array + indexed load)
    __device__ __forceinline__ float sel(float a, float b, float c2, float d) const
    {
        const float lo = (k & 1) ? b : a, hi = (k & 1) ? d : c2;
        return (k & 2) ? hi : lo;
    }
    __device__ __forceinline__ float4 pick(const float4& e0, const float4& e1, const float4& e2, const float4& e3) const
    {
        return make_float4(sel(e0.x, e1.x, e2.x, e3.x), sel(e0.y, e1.y, e2.y, e3.y), sel(e0.z, e1.z, e2.z, e3.z), sel(e0.w, e1.w, e2.w, e3.w));
    }
    __device__ __forceinline__ void head_c(int sub, float q, float v, float w, float wm) const
    {
        if (p) p[(size_t)sub * rs] = make_float4(q, v, w, wm);
    }
    __device__ __forceinline__ void end_c(float pp, float q, float v, float w) const
    {
        if (p) p[(size_t)S * rs] = make_float4(pp, q, v, w);
    }
    // row S + 1 -- what else the adjoint of the step reads of the step's INPUTS, so that it never touches the (HBM-cold) tape slab:
    // (body rates, ring head bits) (angular acceleration, step-counter bits) (the action the interval consumed) before the
    // interval, by lane groups 0..2; (done, d_reward, pre-step gate bits, 0) after the epilogue, by lane group 3
    __device__ __forceinline__ void inputs(const Agent& s, float head_bits, float counter_bits, const float* a) const
    {
        if (p && k < 3)
            p[(size_t)(S + 1) * rs] = pick(make_float4(s.w[0], s.w[1], s.w[2], head_bits), make_float4(s.aa[0], s.aa[1], s.aa[2], counter_bits),
                                           make_float4(a[0], a[1], a[2], a[3]), make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    }
    __device__ __forceinline__ void outcome(bool done, float d_reward, float gate_bits) const
    {
        if (p && k == 3) p[(size_t)(S + 1) * rs] = make_float4(done ? 1.0f : 0.0f, d_reward, gate_bits, 0.0f);
    }
    // row S + 2, entries 0 / 1: the agent's two drag granules (per-agent drag randomisation only)
    __device__ __forceinline__ void drag(const float* S_, int G, int i, int g_drag) const
    {
        if (p && g_drag >= 0 && k < 2) p[(size_t)(S + 2) * rs] = *granule(const_cast<float*>(S_), G, i, g_drag + k);
    }
};

template <class Net, int KIND, int ACT, int INTEG, bool CTRL_DELAY>
__global__ __launch_bounds__(64) void k_bptt_rollout(const vf_dyn_cfg* __restrict__ cp, const vf_env_cfg* __restrict__ ep, const EnvArgs ge,
                                                     const ChainArgs gc, const RollArgs r)
{
    prefetch_kernarg<sizeof(EnvArgs) + sizeof(ChainArgs) + sizeof(RollArgs) + 16>();
    const vf_dyn_cfg& c = *cp;
    const vf_env_cfg& e = *ep;
    constexpr int OW = obs_width(KIND);       // 13, or RacingEnv2's 16 gate-relative columns
    __shared__ __attribute__((aligned(16))) float tile[64 * 13];
    __shared__ float4 act_lds[16];      // the action rows of the step: chain lanes (m, gq = 0) -> the quad of agent slot m
    const int lane = threadIdx.x, m = lane & 15;
    const int wave_first = blockIdx.x * 16;
    // policy rows: lane (m, gq = lane >> 4) works on row m of the wave (MFMA layout).  Env step: the QUAD of lanes 4 m .. 4 m + 3
    // holds agent slot m -- replicas outside the sub-step loop (same index, same loads, same arithmetic, same stores of the same
    // values), the four components of the agent's quantities inside it (vf_dyn_quad.hpp).  Lanes past the last agent replicate it.
    const int i = min(wave_first + m, r.N - 1), ic = min(wave_first + (lane >> 2), r.N - 1);
    const bool live = true;
    EnvArgs g = ge;
    g.d.N = min(r.N, wave_first + 16);                   // the wave's observation tile holds 16 rows
    Agent s;
    Spares sp;
    load_agent<true>(g.d.S, g.d.G, ic, s, sp);
    load_wind(c, g.d, ic, live, s);
    float disc = r.disc[ic], loss = r.loss[ic];
    const QuadLane ql = quad_lane(c, lane);
    // exp(log_std) of the state-independent head once per launch (the head epilogue loaded the four parameters one by one, each an
    // L2 round trip in front of its expf: 2 us per step)
    float std_reg[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (Net::HV != 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) std_reg[k] = expf(gc.rp_log_std[k]);
    }
    const int Gx = g.d.G;
    ChainState16<Net> st;
    chain16_prologue<Net, 0>(gc, st, lane);
    for (int t = 0; t < r.H; ++t) {
        // ---- policy forward + action head: rows t N + i of the slot buffers, action row of step t ----
        {
            // (an opaque copy of the lane id per iteration: the chain's per-item load offsets are loop-invariant and would be
            // hoisted out of the t loop -- dozens of live VGPRs, scratch spills in the two-branch network)
            int lane_t = lane;
            asm volatile("" : "+v"(lane_t));
            const int row = t * r.N + i, rc = row, gq = lane_t >> 4;
            const bool lrow = true;
            // (... and an opaque zero in the weight pointers: the per-item base addresses are loop-invariant too; hoisted, the SGPR
            // pairs are spilled to VGPR lanes and read back with two v_readlane per item)
            long zero_t = 0;
            asm volatile("" : "+s"(zero_t));
            ChainArgs gct = gc;
            gct.packed = gc.packed + zero_t;
            gct.params = gc.params + zero_t;
            gct.rp_std_valid = 1;
#pragma unroll
            for (int k = 0; k < 4; ++k) gct.rp_std[k] = std_reg[k];
            // (the first weight fragments of this step were requested at the end of the previous step's chain, see below)
#pragma unroll
            for (int b = 0; b < Net::NB; ++b) {
                const int w = gct.d.in_dim[b];
                const float* x = gct.io.in[b] + (size_t)rc * w;
                // the state observation of step t > 0 is the row the env epilogue of step t - 1 left in the LDS tile (it also wrote it to
                // slot t: the weight gradients' X); read back from the slot it would be an L2 round trip behind the store
                const float* xl = tile + (lane_t & 15) * OW;
                const bool from_lds = b == 0 && t > 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * gq + j;
                    const float v = from_lds ? xl[k < w ? k : w - 1] : x[k < w ? k : w - 1];
                    st.x[b][j] = k < w ? v : 0.0f;
                }
            }
            chain16_items<Net, 0>(gct, st, lane_t, row, lrow, rc);
            if (gq == 0) act_lds[lane_t & 15] = st.act;
            // the weight fragments the NEXT step's chain starts with (the same every step), requested here, in front of the env step's
            // ~35 stores: vmcnt is ONE in-order counter of loads and stores on gfx9 -- requested at the head of the next chain they
            // would return behind those stores' acknowledgements
            if (t + 1 < r.H) {
                int lane_n = lane;
                asm volatile("" : "+v"(lane_n));
                long zero_n = 0;
                asm volatile("" : "+s"(zero_n));
                ChainArgs gcn = gc;
                gcn.packed = gc.packed + zero_n;
                chain16_prologue<Net, 0>(gcn, st, lane_n);
            }
        }
        // the action row this wave just wrote is what it reads next (other lanes of the SAME wave: program order through the one
        // TCP; a workgroup-scope fence = s_waitcnt only -- an agent-scope __threadfence() adds an L2 write-back per step)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        // ---- checkpoint for the adjoint: this agent's granules of tape row t = the slab before the step ----
        float* T = r.tape + (size_t)t * r.tape_stride;
        store_agent(T, Gx, ic, s, sp);
        // (the other granules -- delay ring, drag, race, spawn copies --, four loads per wait: one by one, every copy is an L2 round
        // trip behind the stores above, 3-4 us per step at 11 granules)
        for (int q0 = VF_G_FIXED; q0 < Gx; q0 += 4) {       // (indices past the last granule repeat it: same value to the same address)
            const int q1 = min(q0 + 1, Gx - 1), q2 = min(q0 + 2, Gx - 1), q3 = min(q0 + 3, Gx - 1);
            const float4 v0 = *granule(g.d.S, Gx, ic, q0), v1 = *granule(g.d.S, Gx, ic, q1), v2 = *granule(g.d.S, Gx, ic, q2),
                         v3 = *granule(g.d.S, Gx, ic, q3);
            *granule(T, Gx, ic, q0) = v0;
            *granule(T, Gx, ic, q1) = v1;
            *granule(T, Gx, ic, q2) = v2;
            *granule(T, Gx, ic, q3) = v3;
        }
        // ---- env step (k_env_rollout's body) ----
        float a[4], head_bits = 0.0f;
        const float head_pre = sp.vel, counter_pre = sp.omg;
        const float4 act_new = act_lds[lane >> 2];       // (LDS operations of a wave execute in order)
        ring_exchange(c, g.d, ic, live, head_bits, a, &act_new);
        if (c.delay_steps > 0) sp.vel = head_bits;
        float kl[3], kq[3];
        drag_of(c, g.d, ic, kl, kq);
        // the noise row of the NEXT step's action head (drawn before the launch: HBM-cold) is touched here, under the dynamics
        // interval; the head's own load at the end of the next forward then finds it in the cache instead of waiting for HBM
        const float4 eps_touch = gc.rp_eps[(size_t)(t + 1 < r.H ? t + 1 : t) * r.N + i];
        // sub-step tape of this (step, wave): [H][S + 3 rows][waves][64] float4 (wave-uniform pointer test: no divergence)
        const size_t ck_rs = (size_t)gridDim.x * 64;
        const TapeCheckpoint ck{r.ck ? r.ck + ((size_t)t * (c.interval_steps + 3) * gridDim.x + blockIdx.x) * 64 + ((lane & 3) * 16 + (lane >> 2)) : nullptr,
                                ck_rs, c.interval_steps, lane & 3};
        ck.inputs(s, head_pre, counter_pre, a);
        ck.drag(g.d.S, Gx, ic, g.d.g_drag);
        float gate_pre = 0.0f;
        if constexpr (kind_is_racing(KIND)) gate_pre = granule(g.d.S, Gx, ic, g.g_race)->x;
        control_interval_quad<ACT, INTEG, CTRL_DELAY>(c, ql, s, a, kl, kq, g.d.vstrided != 0, ck);
        asm volatile("" :: "v"(eps_touch.x), "v"(eps_touch.y), "v"(eps_touch.z), "v"(eps_touch.w));
        float reward = 0.0f;
        bool done = false;
        env_epilogue<KIND, false, 4>(c, e, g, ic, live, s, sp, wave_first, tile, &reward, &done);
        ck.outcome(done, -disc * r.scale, gate_pre);
        if (r.reward_rows) r.reward_rows[(size_t)t * r.N + ic] = reward;
        if (r.ep_flag_rows && done && g.out.ep_flags) r.ep_flag_rows[(size_t)t * r.N + ic] = g.out.ep_flags[ic];   // (this lane's own store above)
        // ---- loss / discount recurrence (BPTT.py:123-124; k_bptt_accumulate) ----
        r.d_reward[(size_t)t * r.N + ic] = -disc * r.scale;
        loss = loss + -1.0f * reward * disc;
        const float dn = done ? 1.0f : 0.0f;
        disc = disc * r.gamma * (1.0f - dn) + dn;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // the observation rows of slot t + 1 are read by the next forward
        // ---- next step ----
        g.d.action += r.N;                               // float4 units
        g.out.done += r.N;
        g.out.obs = t + 2 < r.H ? r.obs_slots + (size_t)(t + 2) * r.N * OW : r.obs_final;   // the last one: the env's own buffer
        g.d.head = g.d.head + 1 == c.delay_steps ? 0 : g.d.head + 1;
    }
    store_agent(g.d.S, Gx, ic, s, sp);
    r.disc[ic] = disc;
    r.loss[ic] = loss;
}

}  // namespace vf

namespace vf {

using RollKernel = void (*)(const vf_dyn_cfg*, const vf_env_cfg*, const vf::EnvArgs, const vf::ChainArgs, const vf::RollArgs);

// DELAY: the motor-lag form of the interval (ctrl_delay, the reference's default: envs/base/dynamics.py:510-533) or the direct one
// (:534-554); the direct instances live in vf_bptt_rollout_nodelay.hip
template <class Net, int KIND, bool DELAY = true>
static RollKernel pick_roll(const vf_dyn_cfg& c)
{
    if ((c.ctrl_delay != 0) != DELAY) return nullptr;
    if (c.integrator == VF_INT_RK4) {
        if (c.action_type == VF_ACT_THRUST) return vf::k_bptt_rollout<Net, KIND, VF_ACT_THRUST, VF_INT_RK4, DELAY>;
        if (c.action_type == VF_ACT_BODYRATE) return vf::k_bptt_rollout<Net, KIND, VF_ACT_BODYRATE, VF_INT_RK4, DELAY>;
        return nullptr;
    }
    if (c.action_type == VF_ACT_THRUST) return vf::k_bptt_rollout<Net, KIND, VF_ACT_THRUST, VF_INT_EULER, DELAY>;
    if (c.action_type == VF_ACT_BODYRATE) return vf::k_bptt_rollout<Net, KIND, VF_ACT_BODYRATE, VF_INT_EULER, DELAY>;
    return nullptr;
}

// vf_bptt_rollout_nodelay.hip: every class below with ctrl_delay = false (cls: chain16_policy_class's 1 .. 4)
RollKernel pick_roll_nodelay(int cls, int kind, const vf_dyn_cfg& c);
// vf_bptt_rollout_nav2.hip: the one-observation classes (1, 3) over the Navigation env kind = NavigationEnv2, whose target is inside its
// "state" row (envs/NavigationEnv.py:163-183); both forms of the interval
RollKernel pick_roll_nav2(int cls, const vf_dyn_cfg& c);

// vf_bptt_rollout_race2.hip: the one-observation classes (1, 3) over RacingEnv2's 16-column rows (kernel-side kind VF_ENV_RACING2)
RollKernel pick_roll_race2(int cls, const vf_dyn_cfg& c);
// vf_bptt_rollout_sac.hip: net = 3 NetSacHover (Hover / Racing env), 4 NetSacNav (Navigation env); nullptr: no instance
RollKernel pick_roll_sac(int net, int kind, const vf_dyn_cfg& c);

}  // namespace vf


namespace vf {

// layout stamp of what a BPTT plugin's roll-out is handed (vf_chain_plugin.hpp: ChainPlugin::bptt_roll_abi)
constexpr unsigned kBpttRollPluginAbi = 0x42520001u ^ (unsigned)(sizeof(EnvArgs) * 31u + sizeof(RollArgs) * 17u + sizeof(vf_dyn_cfg) * 7u +
                                                                sizeof(vf_env_cfg) * 5u + sizeof(ChainArgs) * 3u);

}  // namespace vf

#if defined(VF_CHAIN_PLUGIN) && (VF_CHAIN_PLUGIN_PART == 5 || VF_CHAIN_PLUGIN_PART == 7)
#include "vf_mlp_chain_gen.hpp"
#include "vf_chain_plugin.hpp"
extern "C" int vf_plugin_bptt_rollout(const vf_mlp_desc*, const float*, int, const vf_dyn_cfg*, int, const vf_dyn_cfg*, const vf_env_cfg*, const void*,
                                      const vf::ChainArgs*, const void*, int, hipStream_t);
#endif
#if defined(VF_CHAIN_PLUGIN) && VF_CHAIN_PLUGIN_PART == 5
namespace vf {

// the forward half of a horizon for ONE generated actor class under ONE env kind / action type / integrator / motor-lag setting.  NetR: the
// class the horizon steps -- the SAC-style Actor (both trunks) or the policy-only class of an actor-critic; 16 agents per wave, the
// rows-per-wave choice of the plugin's per-step forward for the row counts this launch serves (vf_chain_plugin.hpp: plugin_forward)
template <class NetR, int KIND, int ACT, int INTEG, bool DELAY>
int plugin_bptt_rollout(const vf_mlp_desc* d, const float* params, int env_kind, const vf_dyn_cfg* c, int has_target, const vf_dyn_cfg* d_dyn,
                        const vf_env_cfg* d_env, const void* env_args, const ChainArgs* gc, const void* roll_args, int N, hipStream_t st)
{
    if (env_kind != KIND || c->action_type != ACT || c->integrator != INTEG || (c->ctrl_delay != 0) != DELAY) return 0;
    if ((NetR::NB == 2) != (has_target != 0) || (NetR::NB == 2 && KIND != VF_ENV_NAV)) return 0;
    if (!chain_matches_gen<NetR>(*d) || !chain16_ok<NetR>(*d, params, 1)) return 0;
    if (NetR::HV == 4 ? (!gc->io.mean || !gc->io.value) : !gc->rp_log_std) return 0;
    hipLaunchKernelGGL((k_bptt_rollout<NetR, KIND, ACT, INTEG, DELAY>), dim3((N + 15) / 16), dim3(64), 0, st, d_dyn, d_env,
                       *static_cast<const EnvArgs*>(env_args), *gc, *static_cast<const RollArgs*>(roll_args));
    VF_HIP(hipGetLastError());
    return 1;
}

}  // namespace vf

#define VF_CHAIN_PLUGIN_BPTT_DEFINE(Net, NetPi, KIND, ACT, INTEG, DELAY, NAME)                                                                  \
    extern "C" int vf_plugin_bptt_rollout(const vf_mlp_desc* d, const float* params, int env_kind, const vf_dyn_cfg* c, int has_target,         \
                                          const vf_dyn_cfg* d_dyn, const vf_env_cfg* d_env, const void* ea, const vf::ChainArgs* gc,            \
                                          const void* ra, int N, hipStream_t st)                                                                \
    {                                                                                                                                           \
        using NetR = std::conditional_t<Net::HV == 4, Net, NetPi>;                                                                              \
        return vf::plugin_bptt_rollout<NetR, KIND, ACT, INTEG, DELAY>(d, params, env_kind, c, has_target, d_dyn, d_env, ea, gc, ra, N, st);     \
    }
#endif
