// vf_env.hip -- the whole visual=False env step in one launch (gfx950).
//
// Replaces, per control step, the chain DroneGymEnvsBase.step -> DroneEnvsBase.step ->
// Dynamics.step -> update_collision -> get_success/get_reward -> done masks -> per-agent
// Python info loop -> examine()/reset_agent_by_id with its per-agent randomizer loop
// (envs/base/droneGymEnv.py:141-218,339-423; envs/base/droneEnv.py:237-288,345-379;
// envs/HoverEnv.py, envs/NavigationEnv.py, envs/RacingEnv.py).  One thread per agent; the
// env counters ride in the spare components of the dynamics granules, so the env step
// moves the same bytes as the bare dynamics step plus its outputs.
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
    // helper blocks REFILL, or -1; helper != 0: the second half of the grid are helper blocks
    int g_spawn_rd = -1, g_spawn_wr = -1, helper = 0;
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

// helper blocks of k_env_step: refill the copy this launch does not read for every agent whose copy is stale
__device__ __forceinline__ void spawn_helper(const vf_env_cfg& e, const EnvArgs& g, int i)
{
    if (i >= g.d.N) return;
    const unsigned need = ((unsigned)__float_as_int(granule(g.d.S, g.d.G, i, VF_G_ACC)->x) >> 8) + 1u;   // episode counter + 1
    float4* dst = granule(g.d.S, g.d.G, i, g.g_spawn_wr);
    if (__float_as_uint(dst->x) == need) return;
    Agent s;
    spawn_agent(e, i, need, true, s);
    const int gs = 64;                                   // float4 between two granules of one agent (wave-tile AoSoA)
    dst[0] = make_float4(__uint_as_float(need), s.p[0], s.p[1], s.p[2]);
    dst[gs] = make_float4(s.q.w, s.q.x, s.q.y, s.q.z);
    dst[2 * gs] = make_float4(s.t, s.v[0], s.v[1], s.v[2]);
    dst[3 * gs] = make_float4(0.0f, s.w[0], s.w[1], s.w[2]);
}

template <int KIND, bool STORE_STATE = true, bool EXT = false>
__device__ __forceinline__ void env_epilogue(const vf_dyn_cfg& c, const vf_env_cfg& e, const EnvArgs& g, int i, bool live,
                                             Agent& s, Spares& sp, int wave_first, float* tile)
{
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

    bool success = false, failure = false;
    float reward;
    int gate = 0, passed = 0, gate_pre = 0;   // gate_pre: the index the terminal observation carries -- the observation is
                                              // refreshed before get_success() advances it (droneGymEnv.py:161-166,197-208)
    float4 race = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (KIND == VF_ENV_HOVER) {
        reward = hover_reward(s.p, e.target, s.q, vel, s.w);
    } else if constexpr (KIND == VF_ENV_NAV) {
        success = norm3(s.p[0] - e.target[0], s.p[1] - e.target[1], s.p[2] - e.target[2]) <= e.success_radius;
        if (e.reward_mode == VF_REWARD_NAV2) {   // NavigationEnv2: failure = is_collision (NavigationEnv.py:159-160)
            failure = col.hit;
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
    er.rewards = er.rewards + reward;                                                    // :185
    bool ep_done = (er.flags & VF_F_EPISODE_DONE) || success || failure || (er.flags & VF_F_OUT_BOUNDS);   // :188
    if (e.is_collision_reset) ep_done = ep_done || (er.flags & VF_F_COLLISION);          // :189-190
    const bool truncated = er.step_count >= e.max_episode_steps;
    const bool done = ep_done || truncated;                                              // :193
    er.flags = set_flag(er.flags, VF_F_EPISODE_DONE, ep_done);
    er.flags = set_flag(er.flags, VF_F_SUCCESS, success);
    er.flags = set_flag(er.flags, VF_F_FAILURE, failure);
    er.flags = set_flag(er.flags, VF_F_DONE, done);

    if (g.out.done_list) {                       // compacted done list: one atomic per wave that has an ending agent
        const unsigned long long m = __ballot(live && done);
        if (m) {
            const int lane = threadIdx.x & 63, first = __ffsll((long long)m) - 1;
            int base = 0;
            if (lane == first) base = atomicAdd(g.out.done_count, __popcll(m));
            base = __shfl(base, first);
            if (live && done) g.out.done_list[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
        }
    }
    // prefetched re-spawn: only a wave that ends an episode touches the copy -- four exec-masked 16-byte loads, issued as soon as
    // `done` is known so that they travel under the terminal-row stores (loading them with the state burst of EVERY wave cost
    // the no-reset launch 0.45 us: profiles/r03_reset_prefetch.txt)
    SpawnSlot slot;
    const bool use_slot = done && g.auto_reset && g.g_spawn_rd >= 0;
    if (use_slot) {
        const float4* src = granule(g.d.S, g.d.G, i, g.g_spawn_rd);
        slot.g0 = src[0]; slot.g1 = src[64]; slot.g2 = src[128]; slot.g3 = src[192];
    }
    float o[13];
    obs_row(c, s, o);
    obs_variant(e, o);
    if (live) {
        st1(g.out.reward + i, reward);
        g.out.done[i] = done ? 1 : 0;
        if (done) {  // collect_info (:238-275)
            if (g.out.ep_return) g.out.ep_return[i] = er.rewards;
            if (g.out.ep_length) g.out.ep_length[i] = er.step_count;
            if (g.out.ep_flags)
                g.out.ep_flags[i] = (success ? VF_EP_SUCCESS : 0) | (truncated ? VF_EP_TRUNCATED : 0) |
                                    ((er.flags & VF_F_ONCE_COLLIDED) ? VF_EP_COLLIDED : 0) |
                                    (ep_done ? VF_EP_EPISODE_DONE : 0);
            if constexpr (KIND == VF_ENV_RACING) {
                if (g.out.ep_past_gates) g.out.ep_past_gates[i] = passed;
                if (g.out.terminal_gate) g.out.terminal_gate[i] = gate_pre;
            }
            if (g.out.terminal_obs) {
                float* to = g.out.terminal_obs + 13 * (size_t)i;
#pragma unroll
                for (int k = 0; k < 13; ++k) to[k] = o[k];
            }
        }
    }
    if (done && g.auto_reset) {  // examine() -> reset_agent_by_id (:339-349,420-423)
        unsigned episode = ((unsigned)er.flags >> 8) + 1u;
        if constexpr (KIND == VF_ENV_RACING) {
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
    if constexpr (KIND == VF_ENV_RACING) {
        race.x = __int_as_float(gate);
        race.y = __int_as_float(passed);
        *granule(g.d.S, g.d.G, i, g.g_race) = race;
        if (live && g.out.gate) g.out.gate[i] = gate;
    }
    pack_env(er, sp);
    if constexpr (STORE_STATE) store_agent(g.d.S, g.d.G, i, s, sp);
    store_rows_coalesced<13>(g.out.obs, g.d.N, wave_first, o, tile);
}

template <int KIND, int ACT, int INTEG, bool CTRL_DELAY>
__global__ __launch_bounds__(kBlock) void k_env_step(const vf_dyn_cfg* __restrict__ cp, const vf_env_cfg* __restrict__ ep, const EnvArgs g)
{
    const vf_dyn_cfg& c = *cp;   // persistent device copies (vf_handles.hpp): L2-resident from launch to launch
    const vf_env_cfg& e = *ep;
    __shared__ __attribute__((aligned(16))) float tile[kBlock * 13];
    if (g.helper) {                                  // second half of the grid: helper blocks (prefetched re-spawn)
        const int nbm = (int)(gridDim.x >> 1);
        if ((int)blockIdx.x >= nbm) {
            spawn_helper(e, g, ((int)blockIdx.x - nbm) * kBlock + (int)threadIdx.x);
            return;
        }
    }
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < g.d.N;
    Agent s;
    Spares sp;
    float a[4], head_bits = 0.0f;
    ring_exchange(c, g.d, i, live, head_bits, a);   // issued first: its two loads are the first values the controller needs
    load_agent<false>(g.d.S, g.d.G, i, s, sp);
    load_wind(c, g.d, i, live, s);
    if (c.delay_steps > 0) sp.vel = head_bits;
    float kl[3], kq[3];
    drag_of(c, g.d, i, kl, kq);
    control_interval<ACT, INTEG, CTRL_DELAY>(c, s, a, kl, kq, g.d.vstrided != 0);
    const int wave = threadIdx.x >> 6;
    env_epilogue<KIND>(c, e, g, i, live, s, sp, blockIdx.x * kBlock + wave * 64, tile + wave * 64 * 13);
}

// The part of the step that follows the dynamics interval, as a launch of its own (vf_env_finish_step): the dynamics ran
// in vf_dyn_step on the same slab, an external scene manager then answered the collision query for the new poses
// (droneEnv.py:374-379 with visual=True).  With ext_point = ext_oob = null the two launches equal one vf_env_step bit for bit.
template <int KIND>
__global__ __launch_bounds__(kBlock) void k_env_finish(const vf_dyn_cfg* __restrict__ cp, const vf_env_cfg* __restrict__ ep, const EnvArgs g)
{
    const vf_dyn_cfg& c = *cp;
    const vf_env_cfg& e = *ep;
    __shared__ __attribute__((aligned(16))) float tile[kBlock * 13];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < g.d.N;
    Agent s;
    Spares sp;
    load_agent<false>(g.d.S, g.d.G, i, s, sp);
    load_wind(c, g.d, i, live, s);
    const int wave = threadIdx.x >> 6;
    env_epilogue<KIND, true, true>(c, e, g, i, live, s, sp, blockIdx.x * kBlock + wave * 64, tile + wave * 64 * 13);
}

// K consecutive steps in one launch: see vf_env_rollout_fused (include/visfly_amd.h)
template <int KIND, int ACT, int INTEG, bool CTRL_DELAY>
__global__ __launch_bounds__(kBlock) void k_env_rollout(const vf_dyn_cfg* __restrict__ cp, const vf_env_cfg* __restrict__ ep, const EnvArgs g0)
{
    const vf_dyn_cfg& c = *cp;
    const vf_env_cfg& e = *ep;
    __shared__ __attribute__((aligned(16))) float tile[kBlock * 13];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    EnvArgs g = g0;
    const bool live = i < g.d.N;
    Agent s;
    Spares sp;
    float a[4], head_bits = 0.0f;
    ring_exchange(c, g.d, i, live, head_bits, a);   // issued first: its two loads are the first values the controller needs
    load_agent<false>(g.d.S, g.d.G, i, s, sp);
    load_wind(c, g.d, i, live, s);
    // one launch = K control steps (vf_env_rollout_fused; a separate kernel from k_env_step: the loop-carried pointers cost the
    // single step ~1 us of register pressure / scheduling when both shared one body).  The agent lives in registers
    // across the steps; the delay ring, per-agent drag and racing granules go through memory as in the single step (a thread
    // sees its own earlier stores), so a reset inside the rollout behaves exactly as between two launches
    for (int k = 0;;) {
        if (c.delay_steps > 0) sp.vel = head_bits;
        float kl[3], kq[3];
        drag_of(c, g.d, i, kl, kq);
        control_interval<ACT, INTEG, CTRL_DELAY>(c, s, a, kl, kq, g.d.vstrided != 0);
        env_epilogue<KIND, false>(c, e, g, i, live, s, sp, blockIdx.x * kBlock + wave * 64, tile + wave * 64 * 13);
        if (++k >= g.K) break;
        g.d.action += g.action_stride;            // float4 units
        g.out.obs += g.obs_stride;
        g.out.reward += g.reward_stride;
        g.out.done += g.done_stride;
        g.d.head = g.d.head + 1 == c.delay_steps ? 0 : g.d.head + 1;
        ring_exchange(c, g.d, i, live, head_bits, a);
    }
    store_agent(g.d.S, g.d.G, i, s, sp);
}

// Two-wave variant (SplitShared in vf_dyn_device.hpp): 256-thread workgroups = 2 rotation + 2 translation
// waves for 128 agents (one wave per SIMD of the CU); the translation waves own the env epilogue and
// every store.
template <int KIND, int ACT, int INTEG, bool CTRL_DELAY>
__global__ __launch_bounds__(kBlock) void k_env_step_split(const vf_dyn_cfg* __restrict__ cp, const vf_env_cfg* __restrict__ ep, const EnvArgs g)
{
    const vf_dyn_cfg& c = *cp;
    const vf_env_cfg& e = *ep;
    __shared__ __attribute__((aligned(16))) SplitShared shs[2];
    const int grp = (threadIdx.x >> 6) & 1;
    SplitShared& sh = shs[grp];
    const int first = blockIdx.x * 128 + grp * 64;
    const int i = first + (threadIdx.x & 63);
    const bool live = i < g.d.N;
    if (threadIdx.x < 128) {
        split_rotation_wave<ACT, INTEG, CTRL_DELAY>(c, g.d, i, live, sh);
        return;
    }
    Agent s;
    Spares sp;
    split_translation_wave<INTEG>(c, g.d, i, sh, s, sp);
    env_epilogue<KIND>(c, e, g, i, live, s, sp, first, sh.tile);
}

struct EnvResetArgs {
    DynArgs d;
    int k, g_race;
    const int* idx;
    const float* fs;  // (k,22) or null
};

template <int KIND>
__global__ __launch_bounds__(kBlock) void k_env_reset(const vf_dyn_cfg c, const vf_env_cfg e, const EnvResetArgs r)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= r.k) return;
    const int i = r.idx ? r.idx[j] : j;
    const bool pad = !r.idx && j >= r.d.N;
    Agent s;
    Spares sp;
    load_agent(r.d.S, r.d.G, i, s, sp);
    EnvRegs er = unpack_env(sp);
    unsigned episode = ((unsigned)er.flags >> 8) + 1u;
    if (!r.idx) { sp.vel = __int_as_float(0); }  // ring head
    const int gate_before = racing_choose_gate(s.p);  // indexed racing resets choose from the OLD position
    reset_rotors(c, s);
    if (r.fs && !pad) {
        const float* f = r.fs + 22 * (size_t)j;
#pragma unroll
        for (int d = 0; d < 3; ++d) { s.p[d] = f[d]; s.v[d] = f[7 + d]; s.w[d] = f[10 + d]; }
        s.q = Quat{f[3], f[4], f[5], f[6]};
#pragma unroll
        for (int d = 0; d < 4; ++d) { s.wm[d] = f[13 + d]; s.T[d] = f[17 + d]; }
        s.t = f[21];
    } else {
        spawn_agent(e, i, episode, r.idx != nullptr, s);
    }
    for (int q = 0; q < c.delay_steps; ++q) *granule(r.d.S, r.d.G, i, VF_G_RING + q) = make_float4(0.f, 0.f, 0.f, 0.f);
    const Collision col = bbox_collision(e, s.p);
    er.flags = (int)(episode << 8);
    er.flags = set_flag(er.flags, VF_F_COLLISION, col.hit);
    er.flags = set_flag(er.flags, VF_F_OUT_BOUNDS, col.oob);
    er.step_count = 0;
    er.rewards = 0.0f;
    pack_env(er, sp);
    store_agent(r.d.S, r.d.G, i, s, sp);
    if (r.d.g_drag >= 0) {
        if (e.drag_random > 0.0f && !(r.fs && !pad)) {  // device spawn: per-agent drag factors
            float4 kl4, kq4;
            spawn_drag(c, e, i, episode, kl4, kq4);
            *granule(r.d.S, r.d.G, i, r.d.g_drag) = kl4;
            *granule(r.d.S, r.d.G, i, r.d.g_drag + 1) = kq4;
        } else if (!r.idx) {  // mean coefficients; a host-side (replay) randomisation overwrites them
            *granule(r.d.S, r.d.G, i, r.d.g_drag) = make_float4(0.f, c.k_lin[0], c.k_lin[1], c.k_lin[2]);
            *granule(r.d.S, r.d.G, i, r.d.g_drag + 1) = make_float4(0.f, c.k_quad[0], c.k_quad[1], c.k_quad[2]);
        }
    }
    if constexpr (KIND == VF_ENV_RACING) {
        float4 race = *granule(r.d.S, r.d.G, i, r.g_race);
        // full reset: _choose_target() runs after the spawn (RacingEnv.py:165-170); indexed: before (:150-163)
        race.x = __int_as_float(r.idx ? gate_before : racing_choose_gate(s.p));
        if (r.idx) race.y = __int_as_float(0);  // RacingEnv.reset keeps _past_targets_num
        race.z = __int_as_float(0);
        *granule(r.d.S, r.d.G, i, r.g_race) = race;
    }
}

__global__ __launch_bounds__(kBlock) void k_env_query(const vf_dyn_cfg c, const vf_env_cfg e, const DynArgs d, int g_race,
                                                      const vf_env_view v)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= d.N) return;
    Agent s;
    Spares sp;
    load_agent(d.S, d.G, i, s, sp);
    const EnvRegs er = unpack_env(sp);
    const Collision col = bbox_collision(e, s.p);
    if (v.step_count) v.step_count[i] = er.step_count;
    if (v.rewards) v.rewards[i] = er.rewards;
    if (v.flags) v.flags[i] = (uint8_t)(er.flags & 0xff);
    if (v.col_dis) v.col_dis[i] = col.dis;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (v.col_point) v.col_point[3 * (size_t)i + k] = col.cp[k];
        if (v.col_vec) v.col_vec[3 * (size_t)i + k] = col.vec[k];
    }
    if (g_race >= 0) {
        const float4 race = *granule(d.S, d.G, i, g_race);
        if (v.gate) v.gate[i] = __float_as_int(race.x);
        if (v.past_gates) v.past_gates[i] = __float_as_int(race.y);
    }
}

// Pose hand-off to an external renderer / scene manager (droneEnv.py:375-377: sceneManager.set_pose(position,
// orientation wxyz, velocity)): AoS rows of the CURRENT slab state, straight from the granules.
__global__ __launch_bounds__(kBlock) void k_env_export_pose(const vf_dyn_cfg c, const DynArgs d, float* pos, float* quat,
                                                            float* vel, float* omg)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= d.N) return;
    const float4 g0 = *granule(d.S, d.G, i, VF_G_POS), g1 = *granule(d.S, d.G, i, VF_G_QUAT);
    const float4 g2 = *granule(d.S, d.G, i, VF_G_VEL), g3 = *granule(d.S, d.G, i, VF_G_OMG);
    if (pos) { float* o = pos + 3 * (size_t)i; o[0] = g0.y; o[1] = g0.z; o[2] = g0.w; }
    if (quat) *(reinterpret_cast<float4*>(quat) + i) = g1;
    if (vel) {   // dynamics.py:751-752
        float4 w = make_float4(c.wind[0], c.wind[1], c.wind[2], 0.0f);
        if (d.wind) w = d.wind[i];
        float* o = vel + 3 * (size_t)i;
        o[0] = g2.y + w.x; o[1] = g2.z + w.y; o[2] = g2.w + w.z;
    }
    if (omg) { float* o = omg + 3 * (size_t)i; o[0] = g3.y; o[1] = g3.z; o[2] = g3.w; }
}

}  // namespace vf

struct vf_env_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    vf_env* env = nullptr;
    int K = 0;
    int phase = 0;   // delay-ring slot of the first captured launch (baked into the kernel arguments)
};

namespace {

using EnvKernel = void (*)(const vf_dyn_cfg*, const vf_env_cfg*, const vf::EnvArgs);

template <int KIND, int ACT>
EnvKernel pick_env_kernel_ka(const vf_dyn_cfg& c)
{
    const int key = (c.integrator == VF_INT_RK4 ? 2 : 0) | (c.ctrl_delay ? 1 : 0);
    switch (key) {
    case 0: return vf::k_env_step<KIND, ACT, VF_INT_EULER, false>;
    case 1: return vf::k_env_step<KIND, ACT, VF_INT_EULER, true>;
    case 2: return vf::k_env_step<KIND, ACT, VF_INT_RK4, false>;
    default: return vf::k_env_step<KIND, ACT, VF_INT_RK4, true>;
    }
}

template <int KIND>
EnvKernel pick_env_kernel_k(const vf_dyn_cfg& c)
{
    switch (c.action_type) {
    case VF_ACT_THRUST: return pick_env_kernel_ka<KIND, VF_ACT_THRUST>(c);
    case VF_ACT_BODYRATE: return pick_env_kernel_ka<KIND, VF_ACT_BODYRATE>(c);
    case VF_ACT_VELOCITY: return pick_env_kernel_ka<KIND, VF_ACT_VELOCITY>(c);
    default: return pick_env_kernel_ka<KIND, VF_ACT_POSITION>(c);
    }
}

template <int KIND, int ACT>
EnvKernel pick_env_rollout_ka(const vf_dyn_cfg& c)
{
    const int key = (c.integrator == VF_INT_RK4 ? 2 : 0) | (c.ctrl_delay ? 1 : 0);
    switch (key) {
    case 0: return vf::k_env_rollout<KIND, ACT, VF_INT_EULER, false>;
    case 1: return vf::k_env_rollout<KIND, ACT, VF_INT_EULER, true>;
    case 2: return vf::k_env_rollout<KIND, ACT, VF_INT_RK4, false>;
    default: return vf::k_env_rollout<KIND, ACT, VF_INT_RK4, true>;
    }
}

template <int KIND>
EnvKernel pick_env_rollout_k(const vf_dyn_cfg& c)
{
    switch (c.action_type) {
    case VF_ACT_THRUST: return pick_env_rollout_ka<KIND, VF_ACT_THRUST>(c);
    case VF_ACT_BODYRATE: return pick_env_rollout_ka<KIND, VF_ACT_BODYRATE>(c);
    case VF_ACT_VELOCITY: return pick_env_rollout_ka<KIND, VF_ACT_VELOCITY>(c);
    default: return pick_env_rollout_ka<KIND, VF_ACT_POSITION>(c);
    }
}

EnvKernel pick_env_rollout(const vf_env* h)
{
    switch (h->cfg.kind) {
    case VF_ENV_HOVER: return pick_env_rollout_k<VF_ENV_HOVER>(h->dyn.cfg);
    case VF_ENV_NAV: return pick_env_rollout_k<VF_ENV_NAV>(h->dyn.cfg);
    default: return pick_env_rollout_k<VF_ENV_RACING>(h->dyn.cfg);
    }
}

template <int KIND>
EnvKernel pick_env_split_k(const vf_dyn_cfg& c)
{
    const int key = (c.action_type == VF_ACT_BODYRATE ? 4 : 0) | (c.integrator == VF_INT_RK4 ? 2 : 0) |
                    (c.ctrl_delay ? 1 : 0);
    switch (key) {
    case 0: return vf::k_env_step_split<KIND, VF_ACT_THRUST, VF_INT_EULER, false>;
    case 1: return vf::k_env_step_split<KIND, VF_ACT_THRUST, VF_INT_EULER, true>;
    case 2: return vf::k_env_step_split<KIND, VF_ACT_THRUST, VF_INT_RK4, false>;
    case 3: return vf::k_env_step_split<KIND, VF_ACT_THRUST, VF_INT_RK4, true>;
    case 4: return vf::k_env_step_split<KIND, VF_ACT_BODYRATE, VF_INT_EULER, false>;
    case 5: return vf::k_env_step_split<KIND, VF_ACT_BODYRATE, VF_INT_EULER, true>;
    case 6: return vf::k_env_step_split<KIND, VF_ACT_BODYRATE, VF_INT_RK4, false>;
    default: return vf::k_env_step_split<KIND, VF_ACT_BODYRATE, VF_INT_RK4, true>;
    }
}

EnvKernel pick_env_split(const vf_env* h)
{
    switch (h->cfg.kind) {
    case VF_ENV_HOVER: return pick_env_split_k<VF_ENV_HOVER>(h->dyn.cfg);
    case VF_ENV_NAV: return pick_env_split_k<VF_ENV_NAV>(h->dyn.cfg);
    default: return pick_env_split_k<VF_ENV_RACING>(h->dyn.cfg);
    }
}

EnvKernel pick_env_kernel(const vf_env* h)
{
    switch (h->cfg.kind) {
    case VF_ENV_HOVER: return pick_env_kernel_k<VF_ENV_HOVER>(h->dyn.cfg);
    case VF_ENV_NAV: return pick_env_kernel_k<VF_ENV_NAV>(h->dyn.cfg);
    default: return pick_env_kernel_k<VF_ENV_RACING>(h->dyn.cfg);
    }
}

vf::DynArgs dyn_args(const vf_env* h, const float* action, float* obs, int ahead = 0)
{
    return vf::DynArgs{h->dyn.N, h->dyn.G, h->dyn.g_drag, h->dyn.S, reinterpret_cast<const float4*>(action), obs,
                       vf::ring_head(&h->dyn, ahead), reinterpret_cast<const float4*>(h->dyn.wind), h->dyn.vel_strided};
}

// `ahead`: position of this launch in a sequence enqueued (or captured) before the handle's step counter advances
int launch_env_step(vf_env* h, const float* action, const vf_env_out* out, int auto_reset, hipStream_t st, int ahead = 0)
{
    vf::EnvArgs g{dyn_args(h, action, out->obs, ahead), *out, h->g_race, auto_reset};
    if (vf::use_split(h->dyn.Npad, h->dyn.cfg)) {
        hipLaunchKernelGGL(pick_env_split(h), dim3(h->dyn.Npad / 128), dim3(vf::kBlock), 0, st, h->dyn.d_cfg, h->d_cfg, g);
    } else {
        unsigned nb = h->dyn.Npad / vf::kBlock;
        if (h->g_spawn >= 0 && auto_reset) {     // prefetched re-spawn: main blocks read copy `par`, helper blocks refill the other
            const int par = (int)((h->dyn.tick + ahead) & 1);
            static const int mode = [] { const char* e = getenv("VISFLY_AMD_PREFETCH_MODE"); return e ? atoi(e) : 3; }();   // A/B: 1 = slot loads only, 2 = helper only
            if (mode & 1) g.g_spawn_rd = h->g_spawn + 4 * par;
            g.g_spawn_wr = h->g_spawn + 4 * (1 - par);
            if (mode & 2) { g.helper = 1; nb *= 2; }
        }
        hipLaunchKernelGGL(pick_env_kernel(h), dim3(nb), dim3(vf::kBlock), 0, st, h->dyn.d_cfg, h->d_cfg, g);
    }
    VF_HIP(hipGetLastError());
    return VF_OK;
}

}  // namespace

extern "C" {

int vf_env_create(const vf_dyn_cfg* dyn, const vf_env_cfg* env, int32_t N, int32_t per_agent_drag, vf_env** out)
{
    if (!dyn || !env || !out || N <= 0) return vf::fail(VF_EINVAL, "vf_env_create: null argument or N <= 0");
    if (int rc = vf::check_dyn_cfg(dyn)) return rc;
    if (env->kind < VF_ENV_HOVER || env->kind > VF_ENV_RACING) return vf::fail(VF_EINVAL, "vf_env_create: bad env kind %d", env->kind);
    if (env->n_spawn < 1 || env->n_spawn > VF_MAX_SPAWN) return vf::fail(VF_EINVAL, "vf_env_create: n_spawn must be 1..%d", VF_MAX_SPAWN);
    if (env->kind == VF_ENV_RACING && (env->n_gates < 1 || env->n_gates > VF_MAX_GATES))
        return vf::fail(VF_EINVAL, "vf_env_create: n_gates must be 1..%d", VF_MAX_GATES);
    if (env->max_episode_steps <= 0) return vf::fail(VF_EINVAL, "vf_env_create: max_episode_steps must be > 0");
    vf_env* h = new vf_env;
    const int racing = env->kind == VF_ENV_RACING ? 1 : 0, extra = racing + (env->spawn_prefetch ? 8 : 0);
    vf::init_dyn_handle(&h->dyn, dyn, N, per_agent_drag, extra);
    h->cfg = *env;
    h->g_race = racing ? h->dyn.g_extra : -1;
    h->g_spawn = env->spawn_prefetch ? h->dyn.g_extra + racing : -1;
    int rc = vf::upload_cfg(h->dyn.cfg, &h->dyn.d_cfg);
    if (rc == VF_OK) rc = vf::upload_cfg(h->cfg, &h->d_cfg);
    if (rc != VF_OK) {
        vf_env_destroy(h);
        return rc;
    }
    *out = h;
    return VF_OK;
}

void vf_env_destroy(vf_env* h)
{
    if (!h) return;
    vf::release_cfg(&h->dyn.d_cfg);
    vf::release_cfg(&h->dyn.d_env_dummy);
    vf::release_cfg(&h->d_cfg);
    delete h;
}

int32_t vf_env_granules(const vf_env* h) { return h ? h->dyn.G : 0; }

int64_t vf_env_slab_floats(const vf_env* h) { return h ? (int64_t)h->dyn.Npad * h->dyn.G * 4 : 0; }

int vf_env_bind(vf_env* h, float* slab)
{
    if (!h) return vf::fail(VF_EINVAL, "vf_env_bind: null handle");
    return vf_dyn_bind(&h->dyn, slab);
}

vf_dyn* vf_env_dyn(vf_env* h) { return h ? &h->dyn : nullptr; }

int vf_env_reset(vf_env* h, const int32_t* idx, int32_t k, const float* full_state, vf_stream_t stream)
{
    if (!h) return vf::fail(VF_EINVAL, "vf_env_reset: null handle");
    if (!h->dyn.S) return vf::fail(VF_ESTATE, "vf_env_reset: vf_env_bind has not been called");
    const int n = idx ? k : h->dyn.Npad;
    if (n < 0) return vf::fail(VF_EINVAL, "vf_env_reset: k < 0");
    if (n == 0) return VF_OK;
    if (!idx) h->dyn.tick = 0;   // full reset: head words go to 0 (k_env_reset) and so does the launch-uniform phase
    if (!idx) h->dyn.vel_strided = 1;   // DroneEnvsBase.reset always passes the randomizer's velocities (droneEnv.py:282)
    vf::EnvResetArgs r{dyn_args(h, nullptr, nullptr), n, h->g_race, idx, full_state};
    hipStream_t st = vf::as_stream(stream);
    const dim3 grid(vf::blocks_for(n)), block(vf::kBlock);
    switch (h->cfg.kind) {
    case VF_ENV_HOVER: hipLaunchKernelGGL(vf::k_env_reset<VF_ENV_HOVER>, grid, block, 0, st, h->dyn.cfg, h->cfg, r); break;
    case VF_ENV_NAV: hipLaunchKernelGGL(vf::k_env_reset<VF_ENV_NAV>, grid, block, 0, st, h->dyn.cfg, h->cfg, r); break;
    default: hipLaunchKernelGGL(vf::k_env_reset<VF_ENV_RACING>, grid, block, 0, st, h->dyn.cfg, h->cfg, r); break;
    }
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_env_step(vf_env* h, const float* action, const vf_env_out* out, int32_t auto_reset, vf_stream_t stream)
{
    if (!h || !action || !out) return vf::fail(VF_EINVAL, "vf_env_step: null argument");
    if (!out->obs || !out->reward || !out->done) return vf::fail(VF_EINVAL, "vf_env_step: obs, reward and done outputs are required");
    if (!h->dyn.S) return vf::fail(VF_ESTATE, "vf_env_step: vf_env_bind has not been called");
    if (int rc = vf::refuse_capture(&h->dyn, vf::as_stream(stream), "vf_env_step")) return rc;
    if ((out->done_list == nullptr) != (out->done_count == nullptr))
        return vf::fail(VF_EINVAL, "vf_env_step: done_list and done_count come together");
    if (out->done_count) VF_HIP(hipMemsetAsync(out->done_count, 0, sizeof(int32_t), vf::as_stream(stream)));
    if (int rc = launch_env_step(h, action, out, auto_reset, vf::as_stream(stream))) return rc;
    h->dyn.tick += 1;
    return VF_OK;
}

namespace {

int check_rollout(const vf_env* h, const vf_env_rollout* r, const char* who)
{
    if (!h || !r) return vf::fail(VF_EINVAL, "%s: null argument", who);
    if (!r->actions || !r->out.obs || !r->out.reward || !r->out.done)
        return vf::fail(VF_EINVAL, "%s: actions, obs, reward and done are required", who);
    if (r->K <= 0) return vf::fail(VF_EINVAL, "%s: K must be > 0", who);
    if (r->action_stride < 0 || r->obs_stride < 0 || r->reward_stride < 0 || r->done_stride < 0)
        return vf::fail(VF_EINVAL, "%s: negative stride", who);
    if (!h->dyn.S) return vf::fail(VF_ESTATE, "%s: vf_env_bind has not been called", who);
    return VF_OK;
}

// the K launches of a rollout, in stream order; step k sees the k-th action / output rows
int enqueue_rollout(vf_env* h, const vf_env_rollout* r, hipStream_t st)
{
    vf_env_out o = r->out;
    o.done_list = o.done_count = nullptr;          // per-step outputs of vf_env_step only
    const float* a = r->actions;
    for (int k = 0; k < r->K; ++k) {
        if (int rc = launch_env_step(h, a, &o, r->auto_reset, st, k)) return rc;
        a += r->action_stride;
        o.obs += r->obs_stride;
        o.reward += r->reward_stride;
        o.done += r->done_stride;
    }
    return VF_OK;
}

}  // namespace

int vf_env_step_n(vf_env* h, const vf_env_rollout* r, vf_stream_t stream)
{
    if (int rc = check_rollout(h, r, "vf_env_step_n")) return rc;
    if (int rc = enqueue_rollout(h, r, vf::as_stream(stream))) return rc;
    h->dyn.tick += r->K;
    return VF_OK;
}

int32_t vf_env_ring_phase(const vf_env* h) { return h ? vf::ring_head(&h->dyn) : 0; }

int vf_env_set_ring_phase(vf_env* h, int32_t phase)
{
    if (!h) return vf::fail(VF_EINVAL, "vf_env_set_ring_phase: null handle");
    return vf::set_ring_phase(&h->dyn, phase, "vf_env_set_ring_phase");
}

int vf_env_rollout_fused(vf_env* h, const vf_env_rollout* r, vf_stream_t stream)
{
    if (int rc = check_rollout(h, r, "vf_env_rollout_fused")) return rc;
    if (r->action_stride % 4) return vf::fail(VF_EINVAL, "vf_env_rollout_fused: action_stride must be a multiple of 4 floats");
    vf::EnvArgs g{dyn_args(h, r->actions, r->out.obs), r->out, h->g_race, r->auto_reset};
    g.out.done_list = g.out.done_count = nullptr;
    g.K = r->K;
    g.action_stride = r->action_stride / 4;
    g.obs_stride = r->obs_stride;
    g.reward_stride = r->reward_stride;
    g.done_stride = r->done_stride;
    hipLaunchKernelGGL(pick_env_rollout(h), dim3(h->dyn.Npad / vf::kBlock), dim3(vf::kBlock), 0, vf::as_stream(stream), h->dyn.d_cfg, h->d_cfg, g);
    VF_HIP(hipGetLastError());
    h->dyn.tick += r->K;
    return VF_OK;
}

int vf_env_graph_create(vf_env* h, const vf_env_rollout* r, vf_env_graph** out)
{
    if (!out) return vf::fail(VF_EINVAL, "vf_env_graph_create: null argument");
    if (int rc = check_rollout(h, r, "vf_env_graph_create")) return rc;
    hipStream_t cs;
    VF_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    vf_env_graph* g = new vf_env_graph;
    g->K = r->K;
    g->env = h;
    g->phase = vf::ring_head(&h->dyn);
    hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed);
    int rc = VF_OK;
    if (e == hipSuccess) {
        rc = enqueue_rollout(h, r, cs);
        e = hipStreamEndCapture(cs, &g->graph);
    }
    if (e == hipSuccess && rc == VF_OK) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    (void)hipStreamDestroy(cs);
    if (e != hipSuccess || rc != VF_OK) {
        if (g->graph) (void)hipGraphDestroy(g->graph);
        delete g;
        if (rc != VF_OK) return rc;
        return vf::fail(VF_EHIP, "vf_env_graph_create: %s", hipGetErrorString(e));
    }
    *out = g;
    return VF_OK;
}

int vf_env_graph_launch(vf_env_graph* g, vf_stream_t stream)
{
    if (!g || !g->exec) return vf::fail(VF_EINVAL, "vf_env_graph_launch: null graph");
    if (vf::ring_head(&g->env->dyn) != g->phase)
        return vf::fail(VF_ESTATE, "vf_env_graph_launch: the graph was captured at delay-ring phase %d, the env is at phase %d "
                                   "(vf_env_ring_phase): capture one graph per phase", g->phase, vf::ring_head(&g->env->dyn));
    VF_HIP(hipGraphLaunch(g->exec, vf::as_stream(stream)));
    g->env->dyn.tick += g->K;
    return VF_OK;
}

void vf_env_graph_destroy(vf_env_graph* g)
{
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}

int vf_env_export_pose(vf_env* h, float* pos, float* quat, float* vel, float* omg, vf_stream_t stream)
{
    if (!h) return vf::fail(VF_EINVAL, "vf_env_export_pose: null handle");
    if (!h->dyn.S) return vf::fail(VF_ESTATE, "vf_env_export_pose: vf_env_bind has not been called");
    if (quat && reinterpret_cast<uintptr_t>(quat) % 16) return vf::fail(VF_EINVAL, "vf_env_export_pose: quat must be 16-byte aligned");
    hipLaunchKernelGGL(vf::k_env_export_pose, dim3(vf::blocks_for(h->dyn.N)), dim3(vf::kBlock), 0, vf::as_stream(stream),
                       h->dyn.cfg, dyn_args(h, nullptr, nullptr), pos, quat, vel, omg);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_env_finish_step(vf_env* h, const float* ext_collision_point, const uint8_t* ext_out_bounds, const vf_env_out* out,
                       int32_t auto_reset, vf_stream_t stream)
{
    if (!h || !out) return vf::fail(VF_EINVAL, "vf_env_finish_step: null argument");
    if (!out->obs || !out->reward || !out->done) return vf::fail(VF_EINVAL, "vf_env_finish_step: obs, reward and done outputs are required");
    if (!h->dyn.S) return vf::fail(VF_ESTATE, "vf_env_finish_step: vf_env_bind has not been called");
    vf::EnvArgs g{dyn_args(h, nullptr, out->obs), *out, h->g_race, auto_reset};
    g.ext_point = ext_collision_point;
    g.ext_oob = ext_out_bounds;
    const dim3 grid(h->dyn.Npad / vf::kBlock), block(vf::kBlock);
    hipStream_t st = vf::as_stream(stream);
    switch (h->cfg.kind) {
    case VF_ENV_HOVER: hipLaunchKernelGGL(vf::k_env_finish<VF_ENV_HOVER>, grid, block, 0, st, h->dyn.d_cfg, h->d_cfg, g); break;
    case VF_ENV_NAV: hipLaunchKernelGGL(vf::k_env_finish<VF_ENV_NAV>, grid, block, 0, st, h->dyn.d_cfg, h->d_cfg, g); break;
    default: hipLaunchKernelGGL(vf::k_env_finish<VF_ENV_RACING>, grid, block, 0, st, h->dyn.d_cfg, h->d_cfg, g); break;
    }
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_env_query(vf_env* h, const vf_env_view* view, vf_stream_t stream)
{
    if (!h || !view) return vf::fail(VF_EINVAL, "vf_env_query: null argument");
    if (!h->dyn.S) return vf::fail(VF_ESTATE, "vf_env_query: vf_env_bind has not been called");
    hipLaunchKernelGGL(vf::k_env_query, dim3(vf::blocks_for(h->dyn.N)), dim3(vf::kBlock), 0, vf::as_stream(stream),
                       h->dyn.cfg, h->cfg, dyn_args(h, nullptr, nullptr), h->g_race, *view);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_env_time_steps(vf_env* h, const float* action, const vf_env_out* out, int32_t auto_reset, int32_t iters,
                      vf_stream_t stream, float* mean_us)
{
    if (!h || !action || !out || !mean_us || iters <= 0) return vf::fail(VF_EINVAL, "vf_env_time_steps: bad argument");
    if (!out->obs || !out->reward || !out->done) return vf::fail(VF_EINVAL, "vf_env_time_steps: obs, reward and done outputs are required");
    if (!h->dyn.S) return vf::fail(VF_ESTATE, "vf_env_time_steps: vf_env_bind has not been called");
    hipStream_t st = vf::as_stream(stream);
    hipEvent_t e0, e1;
    VF_HIP(hipEventCreate(&e0));
    VF_HIP(hipEventCreate(&e1));
    VF_HIP(hipEventRecord(e0, st));
    for (int it = 0; it < iters; ++it) {
        int rc = launch_env_step(h, action, out, auto_reset, st);
        if (rc != VF_OK) return rc;
        h->dyn.tick += 1;
    }
    VF_HIP(hipEventRecord(e1, st));
    VF_HIP(hipEventSynchronize(e1));
    float ms = 0.0f;
    VF_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *mean_us = ms * 1000.0f / (float)iters;
    return VF_OK;
}

}  // extern "C"
