// vf_env_epilogue.hpp -- the per-agent part of the env step that follows the dynamics interval (collision, reward, counters, done
// masks, episode outputs, auto-reset, stores) as device code shared by the step kernels of vf_env.hip and the persistent
// BPTT roll-out of vf_bptt_rollout.hip.
#pragma once
#include "vf_env_device.hpp"
#include "vf_handles.hpp"

#pragma clang fp contract(off)

namespace vf {

struct EnvArgs {
    DynArgs d;
    vf_env_out out;
    int g_race;
    int auto_reset;
    // K consecutive steps inside ONE launch (vf_env_rollout_fused): the agent stays in registers from step to step, only the
    // per-step action is read and the per-step outputs are written; strides in elements between consecutive steps
    int K = 1;
    long long action_stride = 0, obs_stride = 0, reward_stride = 0, done_stride = 0;
    // vf_env_finish_step only: the closest scene point per agent (N,3) and the scene's out-of-bounds flags (N) that an external
    // scene manager computed for the pose the dynamics interval produced (droneEnv.py:330-342); null = the bbox query
    const float* ext_point = nullptr;
    const unsigned char* ext_oob = nullptr;
    // prefetched re-spawn (include/visfly_amd.h): granule index of the copy this launch READS (main waves) and of the copy its
    // helper blocks REFILL, or -1; helper = number of main blocks when helper blocks follow them in the grid, else 0
    int g_spawn_rd = -1, g_spawn_wr = -1, helper = 0;
    // stale bits of the two spawn copies (vf_env::d_stale): [2][n_tiles] words, or null (the helper then checks every agent's tag);
    // stale_wr = index (0 / 1) of the copy the helper blocks refill
    unsigned long long* stale = nullptr;
    int n_tiles = 0, stale_wr = 0;
};

// env counters <-> spare slots
struct EnvRegs {
    int step_count;
    float rewards;
    int flags;  // VF_F_* | episode << 8
};

__device__ __forceinline__ EnvRegs unpack_env(const Spares& sp)
{
    return EnvRegs{__float_as_int(sp.omg), sp.aacc, __float_as_int(sp.acc)};
}
__device__ __forceinline__ void pack_env(const EnvRegs& r, Spares& sp)
{
    sp.omg = __int_as_float(r.step_count);
    sp.aacc = r.rewards;
    sp.acc = __int_as_float(r.flags);
}

__device__ __forceinline__ int set_flag(int flags, int bit, bool on) { return on ? (flags | bit) : (flags & ~bit); }

// collision flags of the current position into the flag word (droneEnv.py:361-369)
__device__ __forceinline__ int collision_flags(int flags, const Collision& col)
{
    flags = set_flag(flags, VF_F_COLLISION, col.hit);
    flags = set_flag(flags, VF_F_OUT_BOUNDS, col.oob);
    if (col.hit) flags |= VF_F_ONCE_COLLIDED;
    return flags;
}

// Everything of DroneGymEnvsBase.step that follows the dynamics interval, for ONE agent held in
// registers: bbox collision, counters, success / reward, done masks, episode outputs, auto-reset,
// stores (envs/base/droneGymEnv.py:161-218,339-423; envs/base/droneEnv.py:345-371).
// one copy of an agent's prefetched re-spawn state (include/visfly_amd.h "Prefetched re-spawn"): (episode tag, p) (q) (t, v) (-, w)
struct SpawnSlot {
    float4 g0, g1, g2, g3;
};

#ifndef VF_HELPER_SPAN
#define VF_HELPER_SPAN 1
#endif
// agents per helper thread.  1: as many helper blocks as main blocks.  Fewer, longer helper blocks were measured and lose: their
// serial refills end up on the launch's critical path (reset regime at 65 536 agents: span 1 11.2 us, 4 11.5, 8 17.6, 16 29.3)
constexpr int kHelperSpan = VF_HELPER_SPAN;

// helper blocks of k_env_step: refill the copy this launch does not read for every agent whose copy is stale.  With the stale bits
// (EnvArgs::stale) a wave reads ONE 64-bit word and is done unless a bit is set; without them it compares every agent's tag.
__device__ __forceinline__ void spawn_helper(const vf_env_cfg& e, const EnvArgs& g, int i)
{
    if (i >= g.d.N) return;
    unsigned long long* word = nullptr;
    if (g.stale) {
        word = g.stale + (size_t)g.stale_wr * g.n_tiles + (i >> 6);
        if (!((*word >> (i & 63)) & 1ull)) return;
    }
    const unsigned need = ((unsigned)__float_as_int(granule(g.d.S, g.d.G, i, VF_G_ACC)->x) >> 8) + 1u;   // episode counter + 1
    float4* dst = granule(g.d.S, g.d.G, i, g.g_spawn_wr);
    if (__float_as_uint(dst->x) != need) {
        Agent s;
        spawn_agent(e, i, need, true, s);
        const int gs = 64;                                   // float4 between two granules of one agent (wave-tile AoSoA)
        dst[0] = make_float4(__uint_as_float(need), s.p[0], s.p[1], s.p[2]);
        dst[gs] = make_float4(s.q.w, s.q.x, s.q.y, s.q.z);
        dst[2 * gs] = make_float4(s.t, s.v[0], s.v[1], s.v[2]);
        dst[3 * gs] = make_float4(0.0f, s.w[0], s.w[1], s.w[2]);
    }
    // this copy now holds the state of episode `need`.  (An agent that ends an episode in THIS launch sets the bit again -- before
    // or after this clear; if before, the bit is lost and the copy looks valid while it is one episode behind: the tag compare at
    // consumption catches it, the in-place draw runs once, and that re-spawn sets the bit again.)
    if (word) atomicAnd(word, ~(1ull << (i & 63)));
}

// the prefetched spawn copy of agent i (granules g_spawn_rd .. + 3 of its tile).  Loaded by EVERY lane, ending an episode or not
// (granule 0 when the feature is off: a valid address whose value nobody looks at): a load under `if (done)` joins a path on which
// the registers are undefined, and the copies the join needs are placed -- with their s_waitcnt -- right behind the loads
// (profiles/r03_reset_prefetch.txt); loads nobody waits for cost an ending-free wave four issue slots
__device__ __forceinline__ SpawnSlot load_spawn_slot(const EnvArgs& g, int i, bool wanted = true)
{
    const float4* src = granule(g.d.S, g.d.G, i, (wanted && g.g_spawn_rd >= 0) ? g.g_spawn_rd : 0);
    SpawnSlot slot;
    slot.g0 = src[0]; slot.g1 = src[64]; slot.g2 = src[128]; slot.g3 = src[192];
    return slot;
}

// LANES = 4: the agent is held by the four lanes of a quad (lanes 4 m .. 4 m + 3 = row m of the wave's 16, k_bptt_rollout): only the
// row hand-over at the end depends on the lane <-> agent map
// (storing what the interval finalises BEFORE collision / reward / counters, with write-through or non-temporal policies, was measured:
// 0.3 us off the headline regime only, profiles/r04_env_quad.txt; not kept)
template <int KIND, bool STORE_STATE = true, int LANES = 1, bool EXT = false, bool LAZY_SLOT = false>
__device__ __forceinline__ void env_epilogue(const vf_dyn_cfg& c, const vf_env_cfg& e, const EnvArgs& g, int i, bool live,
                                             Agent& s, Spares& sp, int wave_first, float* tile, float* reward_reg = nullptr,
                                             bool* done_reg = nullptr, unsigned long long* tr = nullptr)   // tr: -DVF_ENV_TRACE builds only
{
#ifdef VF_ENV_TRACE
#define VF_EPI_TR(k, x) do { if (tr) { asm volatile("" :: "v"(x) : "memory"); tr[k] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define VF_EPI_TR(k, x)
#endif
    EnvRegs er = unpack_env(sp);
    const float vel[3] = {s.v[0] + s.wnd[0], s.v[1] + s.wnd[1], s.v[2] + s.wnd[2]};  // dynamics.py:751-752
    Collision col = bbox_collision(e, s.p);
    if constexpr (EXT) {   // visual branch of update_collision (droneEnv.py:330-342,364-367): the scene manager's closest point
        if (g.ext_point && live) {
            const float* q = g.ext_point + 3 * (size_t)i;
#pragma unroll
            for (int d = 0; d < 3; ++d) { col.cp[d] = q[d]; col.vec[d] = q[d] - s.p[d]; }
            col.dis = norm3(col.vec[0], col.vec[1], col.vec[2]);
            col.hit = col.dis < e.uav_radius;
        }
        if (g.ext_oob && live) col.oob = g.ext_oob[i] != 0;
    }
    er.flags = collision_flags(er.flags, col);
    er.step_count += 1;                                                                  // droneGymEnv.py:163

    // ---- is this the agent's last step?  Decided BEFORE the reward: nothing of it depends on the reward, and a wave that ends
    // an episode wants its prefetched spawn copy on the way as early as possible (below) ----
    bool success = false, failure = false;
    if constexpr (KIND == VF_ENV_NAV) {
        success = norm3(s.p[0] - e.target[0], s.p[1] - e.target[1], s.p[2] - e.target[2]) <= e.success_radius;
        if (e.reward_mode == VF_REWARD_NAV2) failure = col.hit;   // NavigationEnv2: failure = is_collision (NavigationEnv.py:159-160)
    }
    bool ep_done = (er.flags & VF_F_EPISODE_DONE) || success || failure || (er.flags & VF_F_OUT_BOUNDS);   // :188
    if (e.is_collision_reset) ep_done = ep_done || (er.flags & VF_F_COLLISION);          // :189-190
    const bool truncated = er.step_count >= e.max_episode_steps;
    const bool done = ep_done || truncated;                                              // :193
    // prefetched re-spawn: the copy an ending agent starts its next episode from (load_spawn_slot)
    SpawnSlot slot;
    const bool use_slot = done && g.auto_reset && g.g_spawn_rd >= 0;
    if constexpr (!LAZY_SLOT) {
        // Issued by every lane, consumed only by an ending one (a load under `if (done)` is waited for on the spot:
        // profiles/r03_reset_prefetch.txt).  r04: in a wave in which NO agent ends its episode the four loads read the agent's own
        // state granules instead (fetched at the head of the launch: L2 hits) -- a wave-uniform SELECT of the granule index, not a
        // branch, so there is no join whose copies wait for the loads.  r03 read the spawn copy in every wave: 64 B of HBM traffic per
        // agent-step that nobody looks at in the headline regime, where nobody ends an episode (1.25 x the algorithmic bytes)
        const bool wave_ends = __builtin_amdgcn_ballot_w64(use_slot) != 0;
        slot = load_spawn_slot(g, i, wave_ends);
    } else {
        if (use_slot) slot = load_spawn_slot(g, i);
    }
    float reward;
    int gate = 0, passed = 0, gate_pre = 0;   // gate_pre: the index the terminal observation carries -- the observation is
                                              // refreshed before get_success() advances it (droneGymEnv.py:161-166,197-208)
    float4 race = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (KIND == VF_ENV_HOVER) {
        reward = hover_reward(s.p, e.target, s.q, vel, s.w);
    } else if constexpr (KIND == VF_ENV_NAV) {
        if (e.reward_mode == VF_REWARD_NAV2) {
            reward = nav2_reward(e, s.p, vel, s.w, success);
        } else {
            reward = nav_reward(e, s.p, s.q, vel, s.w, col, success, er.step_count, c.trig_mode);
        }
    } else {  // RacingEnv.get_success / get_reward (RacingEnv.py:142-148,199-215)
        race = *granule(g.d.S, g.d.G, i, g.g_race);
        gate = __float_as_int(race.x);
        gate = (unsigned)gate < (unsigned)e.n_gates ? gate : 0;
        gate_pre = gate;
        passed = __float_as_int(race.y);
        const float* gt = e.gates[gate];
        const bool pass = norm3(s.p[0] - gt[0], s.p[1] - gt[1], s.p[2] - gt[2]) <= e.success_radius;
        gate = gate + (pass ? 1 : 0);
        gate = gate == e.n_gates ? 0 : gate;
        passed += pass ? 1 : 0;
        reward = hover_reward(s.p, e.gates[gate], s.q, vel, s.w);
        reward = reward + (pass ? 1.0f : 0.0f) * 20.0f;
        race.z = __int_as_float(pass ? 1 : 0);
    }
    VF_EPI_TR(8, reward);                                                                // collision, done decision, reward
    er.rewards = er.rewards + reward;                                                    // :185
    er.flags = set_flag(er.flags, VF_F_EPISODE_DONE, ep_done);
    er.flags = set_flag(er.flags, VF_F_SUCCESS, success);
    er.flags = set_flag(er.flags, VF_F_FAILURE, failure);
    er.flags = set_flag(er.flags, VF_F_DONE, done);
    if (g.out.done_list) {                       // compacted done list: one atomic per wave that has an ending agent
        const bool mine = live && done && (LANES == 1 || (threadIdx.x & (LANES - 1)) == 0);   // (a quad: its first lane speaks for the agent)
        const unsigned long long m = __ballot(mine);
        if (m) {
            const int lane = threadIdx.x & 63, first = __ffsll((long long)m) - 1;
            int base = 0;
            if (lane == first) base = atomicAdd(g.out.done_count, __popcll(m));
            base = __shfl(base, first);
            if (mine) g.out.done_list[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
        }
    }
    float o[13];
    obs_row(c, s, o);
    obs_variant(e, o);
    if (live) {
        st1(g.out.reward + i, reward);
        if (reward_reg) *reward_reg = reward;       // (callers that keep going with the agent in registers: vf_bptt_rollout.hip)
        if (done_reg) *done_reg = done;
        g.out.done[i] = done ? 1 : 0;
        if (done) {  // collect_info (:238-275)
            if (g.out.ep_return) g.out.ep_return[i] = er.rewards;
            if (g.out.ep_length) g.out.ep_length[i] = er.step_count;
            if (g.out.ep_flags)
                g.out.ep_flags[i] = (success ? VF_EP_SUCCESS : 0) | (truncated ? VF_EP_TRUNCATED : 0) |
                                    ((er.flags & VF_F_ONCE_COLLIDED) ? VF_EP_COLLIDED : 0) |
                                    (ep_done ? VF_EP_EPISODE_DONE : 0);
            if constexpr (kind_is_racing(KIND)) {
                if (g.out.ep_past_gates) g.out.ep_past_gates[i] = passed;
                if (g.out.terminal_gate) g.out.terminal_gate[i] = gate_pre;
            }
            if (g.out.terminal_obs) {
                if constexpr (KIND == VF_ENV_RACING2) {      // the terminal row with the gate index of the step's start (RacingEnv's rule)
                    float ot[16];
                    race2_obs(e, o, gate_pre, ot);
                    float* to = g.out.terminal_obs + 16 * (size_t)i;
#pragma unroll
                    for (int k = 0; k < 16; ++k) to[k] = ot[k];
                } else {
                    float* to = g.out.terminal_obs + 13 * (size_t)i;
#pragma unroll
                    for (int k = 0; k < 13; ++k) to[k] = o[k];
                }
            }
        }
    }
    if (g.stale) {       // both spawn copies of an agent that starts a new episode are one episode behind from now on
        const unsigned long long ending = __builtin_amdgcn_ballot_w64(done && g.auto_reset != 0 && live);
        if (ending && (threadIdx.x & 63) == 0) {
            atomicOr(g.stale + (i >> 6), ending);
            atomicOr(g.stale + g.n_tiles + (i >> 6), ending);
        }
    }
    if (done && g.auto_reset) {  // examine() -> reset_agent_by_id (:339-349,420-423)
        unsigned episode = ((unsigned)er.flags >> 8) + 1u;
        if constexpr (kind_is_racing(KIND)) {
            // RacingEnv.reset_agent_by_id (RacingEnv.py:150-163) picks the next gate BEFORE the base class
            // re-spawns the agent: the choice is made from the terminal position of the finished episode
            gate = racing_choose_gate(s.p);
            passed = 0;
            race.z = __int_as_float(0);
        }
        if (use_slot && __float_as_uint(slot.g0.x) == episode) {   // the state this episode starts from was drawn ahead of time
            s.p[0] = slot.g0.y; s.p[1] = slot.g0.z; s.p[2] = slot.g0.w;
            s.q = Quat{slot.g1.x, slot.g1.y, slot.g1.z, slot.g1.w};
            s.t = slot.g2.x;
            s.v[0] = slot.g2.y; s.v[1] = slot.g2.z; s.v[2] = slot.g2.w;
            s.w[0] = slot.g3.y; s.w[1] = slot.g3.z; s.w[2] = slot.g3.w;
        } else {
            spawn_agent(e, i, episode, true, s);
        }
        reset_rotors(c, s);
        for (int q = 0; q < c.delay_steps; ++q)
            *granule(g.d.S, g.d.G, i, VF_G_RING + q) = make_float4(0.f, 0.f, 0.f, 0.f);     // dynamics.py:262-263
        if (g.d.g_drag >= 0 && e.drag_random > 0.0f) {
            float4 kl4, kq4;
            spawn_drag(c, e, i, episode, kl4, kq4);
            *granule(g.d.S, g.d.G, i, g.d.g_drag) = kl4;
            *granule(g.d.S, g.d.G, i, g.d.g_drag + 1) = kq4;
        }
        col = bbox_collision(e, s.p);                                                       // droneEnv.py:285-288
        er.flags = (int)(episode << 8);
        er.flags = set_flag(er.flags, VF_F_COLLISION, col.hit);
        er.flags = set_flag(er.flags, VF_F_OUT_BOUNDS, col.oob);
        er.step_count = 0;                                                                  // :387-392
        er.rewards = 0.0f;
        obs_row(c, s, o);
        obs_variant(e, o);
    }
    if constexpr (kind_is_racing(KIND)) {
        race.x = __int_as_float(gate);
        race.y = __int_as_float(passed);
        *granule(g.d.S, g.d.G, i, g.g_race) = race;
        if (live && g.out.gate) g.out.gate[i] = gate;
    }
    pack_env(er, sp);
    VF_EPI_TR(9, sp.acc);                                                                // outputs written, re-spawn decided
    if constexpr (STORE_STATE) store_agent(g.d.S, g.d.G, i, s, sp);
    VF_EPI_TR(10, sp.acc);                                                               // state stores issued
    if constexpr (KIND == VF_ENV_RACING2) {          // the row the policy reads next: the CURRENT gate (after a pass / a re-spawn)
        float o16[16];
        race2_obs(e, o, gate, o16);
        if constexpr (LANES == 4) store_rows_quads<16>(g.out.obs, g.d.N, wave_first, o16, tile);
        else store_rows_coalesced<16>(g.out.obs, g.d.N, wave_first, o16, tile);
    } else if constexpr (LANES == 4) store_rows_quads<13>(g.out.obs, g.d.N, wave_first, o, tile);
    else store_rows_coalesced<13>(g.out.obs, g.d.N, wave_first, o, tile);
}

}  // namespace vf
