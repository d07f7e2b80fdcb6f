// vf_ppo_rollout.hip -- PPO's collect_rollouts as ONE persistent launch (gfx950).
//
// SB3 OnPolicyAlgorithm.collect_rollouts (run by utils/algorithms/PPO.py:146) is the closed loop policy.forward(obs_t) ->
// distribution.sample / log_prob -> env.step -> RolloutBuffer.add, n_steps times.  Launch by launch that is, per control step
// at 32 768 agents: the register-chained forward (30 us), the head sampler (5 us), two observation copies (2 x 5 us), the env
// step (11 us) and the TimeLimit bookkeeping (4 us) -- 60 us, of which the launch boundaries and the cold first touches of
// every launch are a third (profiles/r03_ppo_kernel_stats.txt).  Here, as in vf_bptt_rollout.hip, a wave owns its agents for
// all n_steps:
//   * ROWS = 32 agents per wave with the 32-row chain (v_mfma_f32_32x32x2_f32) or 16 with the 16-row chain -- the SAME choice
//     vf_mlp_forward makes from the row count (chain16_ok), so heads, values, actions and log-probs are the per-step path's
//     to the bit.  Lanes ROWS..63 replicate lane & (ROWS - 1) outside the chain (k_bptt_rollout's construction);
//   * nothing of a step travels through memory: the accumulator lane that holds the heads of row m IS lane m (replicas take
//     them by ds_bpermute), the sampled action feeds the env step in registers, the reward / done of the epilogue feed the buffer
//     rows and the TimeLimit list, and the next forward reads its "state" row from the LDS tile the epilogue staged it through
//     on its way to RolloutBuffer.obs["state"][t + 1].  Global memory sees only the buffer rows, written once, never re-read.
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
    float* obs_slots;               // [T][N][13] = RolloutBuffer.obs["state"]
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
template <class Net, int ROWS>
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
            const float* x = b == 0 ? tile + m * 13 : gc.io.in[b] + (size_t)row * w;
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
            const float* x = b == 0 ? tile + m * 13 : gc.io.in[b] + (size_t)row * w;
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
    __shared__ __attribute__((aligned(16))) float tile[64 * 13];
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
    for (int k = 0; k < 13; ++k) tile[lane * 13 + k] = r.obs_slots[(size_t)i * 13 + k];
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
        float4 mean = policy_rows<Net, ROWS>(gct, lane_t, row, tile);
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
                    const float* to = g.out.terminal_obs + 13 * (size_t)i;
                    for (int k = 0; k < 13; ++k) r.rows0[(size_t)slot * 13 + k] = to[k];
                    for (int k = 0; k < r.w1; ++k) r.rows1[(size_t)slot * r.w1 + k] = r.obs1[(size_t)i * r.w1 + k];
                }
            }
        }
        g.out.obs = t + 2 < r.T ? r.obs_slots + (size_t)(t + 2) * r.N * 13 : r.obs_final;
        g.d.head = g.d.head + 1 == c.delay_steps ? 0 : g.d.head + 1;
        VF_PT(4);
    }
    store_agent(g.d.S, Gx, i, s, sp);
#ifdef VF_PPO_TRACE
    if (blockIdx.x == 7 && lane == 0) for (int k = 0; k < 5; ++k) vf_ppo_trace[k] = tr[k];
#endif
}

}  // namespace vf

namespace {

using PpoRollKernel = void (*)(const vf_dyn_cfg*, const vf_env_cfg*, const vf::EnvArgs, const vf::ChainArgs, const vf::PpoRollArgs);

template <class Net, int ROWS, int KIND, bool DELAY>
PpoRollKernel pick_ppo_roll2(const vf_dyn_cfg& c)
{
    if (c.integrator == VF_INT_RK4) {      // BASELINE configs[2]'s dynamics (utils/maths.py:353-386, repaired): bodyrate; thrust too
        if (c.action_type == VF_ACT_THRUST) return vf::k_ppo_rollout<Net, ROWS, KIND, VF_ACT_THRUST, VF_INT_RK4, DELAY>;
        if (c.action_type == VF_ACT_BODYRATE) return vf::k_ppo_rollout<Net, ROWS, KIND, VF_ACT_BODYRATE, VF_INT_RK4, DELAY>;
        return nullptr;
    }
    if (c.action_type == VF_ACT_THRUST) return vf::k_ppo_rollout<Net, ROWS, KIND, VF_ACT_THRUST, VF_INT_EULER, DELAY>;
    if (c.action_type == VF_ACT_BODYRATE) return vf::k_ppo_rollout<Net, ROWS, KIND, VF_ACT_BODYRATE, VF_INT_EULER, DELAY>;
    return nullptr;
}

// motor lag (ctrl_delay, the reference's default) or the direct form of the interval (envs/base/dynamics.py:510-554; r05)
template <class Net, int ROWS, int KIND>
PpoRollKernel pick_ppo_roll(const vf_dyn_cfg& c)
{
    return c.ctrl_delay ? pick_ppo_roll2<Net, ROWS, KIND, true>(c) : pick_ppo_roll2<Net, ROWS, KIND, false>(c);
}

}  // namespace

extern "C" int vf_ppo_rollout(vf_env* h, const vf_mlp_desc* desc, const float* params, const float* packed, const vf_ppo_rollout_args* a,
                              vf_stream_t stream)
{
    if (!h || !desc || !params || !packed || !a || !a->out || !a->obs_state || !a->values || !a->actions || !a->log_probs || !a->rewards ||
        !a->episode_starts || !a->last_starts || !a->obs_final || !a->log_std || !a->cursor || !a->idx_list || !a->rows0 ||
        !a->stat || a->T <= 0 || a->w1 < 0 || (a->w1 > 0 && (!a->rows1 || !a->obs_target_row)))
        return vf::fail(VF_EINVAL, "vf_ppo_rollout: bad argument");
    const vf_env_out* out = a->out;
    if (!out->reward || !out->done || !out->ep_return || !out->ep_length || !out->ep_flags || !out->terminal_obs)
        return vf::fail(VF_EINVAL, "vf_ppo_rollout: out needs reward / done scratch and the episode outputs");
    for (int l = 0; l < desc->n_layers; ++l)
        if (desc->layer[l].save) return vf::fail(VF_EINVAL, "vf_ppo_rollout: the layer table must not keep activation copies (layer %d has a save pointer)", l);
    if (!h->dyn.S) return vf::fail(VF_ESTATE, "vf_ppo_rollout: vf_env_bind has not been called");
    if (h->dyn.wind) return vf::fail(VF_EUNSUPPORTED, "vf_ppo_rollout: per-agent wind rows are set");
    // (observation / reward variants -- HoverEnv2, NavigationEnv2 -- are the epilogue's own: the rows it stages through the wave's
    // tile for the next forward are already in the env's obs_mode, the terminal rows too; r05)
    if ((reinterpret_cast<uintptr_t>(a->means) | reinterpret_cast<uintptr_t>(a->actions) | reinterpret_cast<uintptr_t>(a->stat)) & 15)
        return vf::fail(VF_EINVAL, "vf_ppo_rollout: means / actions / stat must be 16-byte aligned");
    const int N = h->dyn.N, T = a->T;
    // the rows-per-wave choice of vf_mlp_forward for N rows (chain16_ok), so that the heads are the per-step path's to the bit
    const int cls = vf::chain_full_class(desc, params, N);        // 0 none; 1 NetHover, 2 NetNav; + 16 when the 16-row chain runs N rows
    const bool r16 = (cls & 16) != 0;
    PpoRollKernel k = nullptr;
    if ((cls & 15) == 1 && h->cfg.kind == VF_ENV_HOVER)
        k = r16 ? pick_ppo_roll<vf::NetHover, 16, VF_ENV_HOVER>(h->dyn.cfg) : pick_ppo_roll<vf::NetHover, 32, VF_ENV_HOVER>(h->dyn.cfg);
    else if ((cls & 15) == 2 && h->cfg.kind == VF_ENV_NAV && a->obs_target)
        k = r16 ? pick_ppo_roll<vf::NetNav, 16, VF_ENV_NAV>(h->dyn.cfg) : pick_ppo_roll<vf::NetNav, 32, VF_ENV_NAV>(h->dyn.cfg);
    else if ((cls & 15) == 1 && h->cfg.kind == VF_ENV_NAV && !a->obs_target)      // NavigationEnv2: the target is inside the "state" row
        k = r16 ? pick_ppo_roll<vf::NetHover, 16, VF_ENV_NAV>(h->dyn.cfg) : pick_ppo_roll<vf::NetHover, 32, VF_ENV_NAV>(h->dyn.cfg);
    if (!k) return vf::fail(VF_EUNSUPPORTED, "vf_ppo_rollout: no persistent roll-out for this network class / env kind / dynamics "
                                             "configuration ([128, 64] x [64, 64] actor-critic, Hover / Navigation, thrust / bodyrate, Euler / RK4)");
    const int rows = r16 ? 16 : 32;
    vf::EnvArgs ge{vf::DynArgs{N, h->dyn.G, h->dyn.g_drag, h->dyn.S, nullptr, nullptr, vf::ring_head(&h->dyn), nullptr, h->dyn.vel_strided},
                   *out, h->g_race, 1};
    ge.out.done_list = ge.out.done_count = nullptr;
    ge.out.obs = T > 1 ? a->obs_state + (size_t)N * 13 : a->obs_final;
    vf::ChainArgs gc{*desc, params, packed, vf::ChainIo{{a->obs_state, a->obs_target}, a->means, a->values}, T * N, nullptr, nullptr, nullptr,
                     {nullptr, nullptr}};
    vf::PpoRollArgs r{T, N, reinterpret_cast<float4*>(a->actions), a->log_probs, a->rewards, a->episode_starts, a->last_starts,
                      a->obs_state, a->obs_final, a->log_std, a->noise_key, a->sample_step, a->obs_target_row, a->w1, a->capacity, a->cursor,
                      a->idx_list, a->rows0, a->rows1, a->stat};
    hipLaunchKernelGGL(k, dim3((N + rows - 1) / rows), dim3(64), 0, vf::as_stream(stream), h->dyn.d_cfg, h->d_cfg, ge, gc, r);
    VF_HIP(hipGetLastError());
    h->dyn.tick += T;
    h->stale_all = 1;       // agents re-spawned inside the launch: the prefetched copies' stale bits no longer cover them
    return VF_OK;
}

#ifdef VF_PPO_TRACE
extern "C" int vf_debug_ppo_trace(long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(vf::vf_ppo_trace), sizeof(long long) * 8, 0, hipMemcpyDeviceToHost);
}
#endif
