// vf_ppo_rollout_kernel.hpp -- the persistent PPO roll-out kernel as a template over the network class (scheme: vf_ppo_rollout.hip), shared
// by vf_ppo_rollout.hip (the built-in classes) and the chain plugins of generated classes (vf_chain_plugin.hpp, part 4).
#pragma once
#include "vf_env_epilogue.hpp"
#include "vf_mlp_chain.hpp"

#pragma clang fp contract(off)

namespace vf {

#ifdef VF_PPO_TRACE
__device__ long long vf_ppo_trace[8];
#endif

struct PpoRollArgs {
    int T, N;
    float4* actions;                // [T][N]
    float* log_probs;               // [T][N]
    float* rewards;                 // [T][N]
    float* episode_starts;          // [T][N]; row 0 is filled by the caller
    float* last_starts;             // (N,): episode_starts of the step after the last one
    float* obs_slots;               // [T][N][13] = RolloutBuffer.obs["state"] (RacingEnv2: 16 wide, like obs_final / rows0 / the terminal rows)
    float* obs_final;               // (N,13)
    const float* log_std;
    unsigned long long noise_key, sample_step;      // step t samples with Philox counter sample_step + 1 + t (k_head_sample)
    // deferred TimeLimit bootstrap list + per-agent episode statistics (k_rollout_post_collect)
    const float* obs1;              // (N,w1) constant "target" rows or null
    int w1, capacity;
    int* cursor;
    int* idx_list;
    float* rows0;                   // [capacity][13]
    float* rows1;                   // [capacity][w1]
    float* stat;                    // (N,4)
};

// one forward of rows `row` (lane & (ROWS - 1) of the wave): value -> g.io.value; -> the mean row, valid in the lanes < ROWS
// (the accumulator lanes of group 0 hold heads of their own row: chain_epilogue / chain16_epilogue).  The "state" row comes from
// the wave's LDS tile (13 floats per agent, where the env epilogue of the previous step left it), other branches from memory.
template <class Net, int ROWS, int OW = 13>
__device__ __forceinline__ float4 policy_rows(const ChainArgs& gc, int lane, int row, const float* tile)
{
    const int m = lane & (ROWS - 1);
    if constexpr (ROWS == 16) {
        const int gq = lane >> 4;
        ChainState16<Net> st;
        chain16_prologue<Net, 0>(gc, st, lane);
#pragma unroll
        for (int b = 0; b < Net::NB; ++b) {
            const int w = gc.d.in_dim[b];
            const float* x = b == 0 ? tile + m * OW : gc.io.in[b] + (size_t)row * w;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * gq + j;
                const float v = x[k < w ? k : w - 1];
                st.x[b][j] = k < w ? v : 0.0f;
            }
        }
        chain16_items<Net, 0, false>(gc, st, lane, row, true, row);
        const f32x4& y = st.t[2 * Net::t_mean];
        return make_float4(y[0], y[1], y[2], y[3]);
    } else {
        const int h = lane >> 5;
        ChainState<Net> st;
        chain_prologue<Net, 0>(gc, st, lane);
#pragma unroll
        for (int b = 0; b < Net::NB; ++b) {
            const int w = gc.d.in_dim[b];
            const float* x = b == 0 ? tile + m * OW : gc.io.in[b] + (size_t)row * w;
#pragma unroll
            for (int s = 0; s < Net::kin(b) / 2; ++s) {
                const int k = 2 * s + h;
                const float v = x[k < w ? k : w - 1];
                st.x[b][s] = k < w ? v : 0.0f;
            }
        }
        chain_items<Net, 0, false>(gc, st, lane, row, true, row);
        const f32x16& y = st.t[Net::t_mean];
        return make_float4(y[0], y[1], y[2], y[3]);
    }
}

template <class Net, int ROWS, int KIND, int ACT, int INTEG, bool CTRL_DELAY>
__global__ __launch_bounds__(64) void k_ppo_rollout(const vf_dyn_cfg* __restrict__ cp, const vf_env_cfg* __restrict__ ep, const EnvArgs ge,
                                                    const ChainArgs gc, const PpoRollArgs r)
{
    prefetch_kernarg<sizeof(EnvArgs) + sizeof(ChainArgs) + sizeof(PpoRollArgs) + 16>();
    const vf_dyn_cfg& c = *cp;
    const vf_env_cfg& e = *ep;
    constexpr int OW = obs_width(KIND);       // 13, or RacingEnv2's 16 gate-relative columns (vf_env_device.hpp: race2_obs)
    __shared__ __attribute__((aligned(16))) float tile[64 * OW];
    const int lane = threadIdx.x, m = lane & (ROWS - 1);
    const int wave_first = blockIdx.x * ROWS;
    // lanes ROWS..63 and the lanes past the last agent are REPLICAS of a live lane (same index, loads, arithmetic, stores of the
    // same values); what must happen once per agent -- the read-modify-write of the statistics, the list append -- is the owner's
    const int i = min(wave_first + m, r.N - 1);
    const bool owner = lane < ROWS && wave_first + lane < r.N;
    EnvArgs g = ge;
    g.d.N = min(r.N, wave_first + ROWS);                  // the wave's observation tile holds ROWS rows
    Agent s;
    Spares sp;
    load_agent<true>(g.d.S, g.d.G, i, s, sp);
    load_wind(c, g.d, i, true, s);
    const int Gx = g.d.G;
    // the wave's observation tile: row l = the "state" observation of lane l's agent.  The env epilogue of step t leaves the rows
    // of step t + 1 there (store_rows_coalesced stages them through it on their way to RolloutBuffer.obs[t + 1]); step 0's come
    // from the caller's row 0.  One wave's LDS operations execute in order: no barrier.
    for (int k = 0; k < OW; ++k) tile[lane * OW + k] = r.obs_slots[(size_t)i * OW + k];
    __builtin_amdgcn_wave_barrier();
#ifdef VF_PPO_TRACE
    long long tr[5] = {0, 0, 0, 0, 0}, tc = __builtin_readcyclecounter();
#define VF_PT(k) do { const long long n_ = __builtin_readcyclecounter(); tr[k] += n_ - tc; tc = n_; } while (0)
#else
#define VF_PT(k) do { } while (0)
#endif
    for (int t = 0; t < r.T; ++t) {
        const int row = t * r.N + i;
        // the chain's per-item load offsets (lane * 16 + item * 1 KiB) are loop-invariant: hoisted out of the t loop they are
        // ~100 live VGPRs and 1.1 KB of scratch per lane.  An opaque copy of the lane id per iteration keeps them just-in-time
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        // the delay-ring slot this step swaps its action with: address known now, value needed right after the sampler -- loaded
        // ahead of the forward (by every lane, from a valid granule when there is no ring: a load under `if` would be waited for
        // on the spot, see load_spawn_slot)
        const float4 ring_old = *granule(g.d.S, Gx, i, c.delay_steps > 0 ? VF_G_RING + g.d.head : 0);
        // ... and an opaque zero in the weight pointers: the ~170 per-item base addresses (SGPR pairs) are loop-invariant as
        // well, hoisted they are spilled to VGPR lanes and cost two v_readlane per item
        long zero_t = 0;
        asm volatile("" : "+s"(zero_t));
        ChainArgs gct = gc;
        gct.packed = gc.packed + zero_t;
        gct.params = gc.params + zero_t;
        float4 mean = policy_rows<Net, ROWS, OW>(gct, lane_t, row, tile);
        // lane m < ROWS holds the head of its own agent; the replica lanes take it from there
        mean.x = __shfl(mean.x, m); mean.y = __shfl(mean.y, m); mean.z = __shfl(mean.z, m); mean.w = __shfl(mean.w, m);
        VF_PT(0);
        float4 act;
        const float lp = head_sample_row(mean, r.log_std, i, r.noise_key, r.sample_step + 1ull + (unsigned long long)t, 0, act);
        r.actions[row] = act;
        r.log_probs[row] = lp;
        // ---- env step (k_env_rollout's body; ring_exchange with the action already in registers) ----
        float a[4];
        {
            float4 an = act;
            if (c.delay_steps > 0) {
                const int head = g.d.head;
                st4(granule(g.d.S, Gx, i, VF_G_RING + head), an);
                an = ring_old;
                sp.vel = __int_as_float(head + 1 == c.delay_steps ? 0 : head + 1);
            }
            a[0] = an.x; a[1] = an.y; a[2] = an.z; a[3] = an.w;
        }
        float kl[3], kq[3];
        drag_of(c, g.d, i, kl, kq);
        VF_PT(1);
        control_interval<ACT, INTEG, CTRL_DELAY>(c, s, a, kl, kq, g.d.vstrided != 0);
        VF_PT(2);
        float reward = 0.0f;
        bool done = false;
        env_epilogue<KIND, false>(c, e, g, i, true, s, sp, wave_first, tile, &reward, &done);
        VF_PT(3);
        // ---- RolloutBuffer.add + the TimeLimit bookkeeping (k_rollout_post_collect) ----
        r.rewards[row] = reward;
        (t + 1 < r.T ? r.episode_starts + (size_t)(t + 1) * r.N : r.last_starts)[i] = done ? 1.0f : 0.0f;
        if (done && owner) {
            // ep_return / ep_length / ep_flags / terminal row: this lane's own stores of the epilogue
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            const unsigned char fl = g.out.ep_flags[i];
            float4* sa = reinterpret_cast<float4*>(r.stat) + i;
            float4 v = *sa;
            v.x += 1.0f;
            v.y += g.out.ep_return[i];
            v.z += (float)g.out.ep_length[i];
            v.w += (fl & VF_EP_SUCCESS) ? 1.0f : 0.0f;
            *sa = v;
            if (fl & VF_EP_TRUNCATED) {
                const int slot = atomicAdd(r.cursor, 1);
                if (slot < r.capacity) {
                    r.idx_list[slot] = row;
                    const float* to = g.out.terminal_obs + OW * (size_t)i;
                    for (int k = 0; k < OW; ++k) r.rows0[(size_t)slot * OW + k] = to[k];
                    for (int k = 0; k < r.w1; ++k) r.rows1[(size_t)slot * r.w1 + k] = r.obs1[(size_t)i * r.w1 + k];
                }
            }
        }
        g.out.obs = t + 2 < r.T ? r.obs_slots + (size_t)(t + 2) * r.N * OW : r.obs_final;
        g.d.head = g.d.head + 1 == c.delay_steps ? 0 : g.d.head + 1;
        VF_PT(4);
    }
    store_agent(g.d.S, Gx, i, s, sp);
#ifdef VF_PPO_TRACE
    if (blockIdx.x == 7 && lane == 0) for (int k = 0; k < 5; ++k) vf_ppo_trace[k] = tr[k];
#endif
}

}  // namespace vf

namespace vf {

// layout stamp of what a roll-out plugin is handed (vf_chain_plugin.hpp: ChainPlugin::rollout_abi)
constexpr unsigned kRolloutPluginAbi = 0x52300001u ^ (unsigned)(sizeof(EnvArgs) * 31u + sizeof(PpoRollArgs) * 17u + sizeof(vf_dyn_cfg) * 7u +
                                                               sizeof(vf_env_cfg) * 5u + sizeof(ChainArgs) * 3u);

}  // namespace vf

#if defined(VF_CHAIN_PLUGIN) && VF_CHAIN_PLUGIN_PART == 4
#include "vf_mlp_chain_gen.hpp"
#include "vf_chain_plugin.hpp"
namespace vf {

// the persistent roll-out of ONE generated class under ONE env kind / action type / integrator / motor-lag setting, 32 rows per wave
// (the rows-per-wave choice of the class's per-step forward at every row count, so that heads / values / log-probs are the loop's)
template <class Net, int KIND, int ACT, int INTEG, bool DELAY>
int plugin_ppo_rollout(const vf_mlp_desc* d, int env_kind, const vf_dyn_cfg* c, int has_target, const vf_dyn_cfg* d_dyn, const vf_env_cfg* d_env,
                       const void* env_args, const ChainArgs* gc, const void* roll_args, int N, hipStream_t st)
{
    if (env_kind != KIND || c->action_type != ACT || c->integrator != INTEG || (c->ctrl_delay != 0) != DELAY) return 0;
    if ((Net::NB == 2) != (has_target != 0) || (Net::NB == 2 && KIND != VF_ENV_NAV)) return 0;
    if (!chain_matches_gen<Net>(*d)) return 0;
    hipLaunchKernelGGL((k_ppo_rollout<Net, 32, KIND, ACT, INTEG, DELAY>), dim3((N + 31) / 32), dim3(64), 0, st, d_dyn, d_env,
                       *static_cast<const EnvArgs*>(env_args), *gc, *static_cast<const PpoRollArgs*>(roll_args));
    VF_HIP(hipGetLastError());
    return 1;
}

}  // namespace vf

#define VF_CHAIN_PLUGIN_ROLLOUT_DEFINE(Net, KIND, ACT, INTEG, DELAY, NAME)                                                                   \
    static int vf_plugin_ppo_rollout(const vf_mlp_desc* d, int env_kind, const vf_dyn_cfg* c, int has_target, const vf_dyn_cfg* d_dyn,       \
                                     const vf_env_cfg* d_env, const void* ea, const vf::ChainArgs* gc, const void* ra, int N, hipStream_t st) \
    { return vf::plugin_ppo_rollout<Net, KIND, ACT, INTEG, DELAY>(d, env_kind, c, has_target, d_dyn, d_env, ea, gc, ra, N, st); }             \
    extern "C" const vf::ChainPlugin* vf_chain_plugin()                                                                                      \
    {                                                                                                                                        \
        static const vf::ChainPlugin p{vf::kChainPluginAbi, NAME, nullptr, nullptr, nullptr, nullptr, vf::kRolloutPluginAbi, vf_plugin_ppo_rollout, 0u, 0u, nullptr, nullptr};  \
        return &p;                                                                                                                           \
    }
#endif
