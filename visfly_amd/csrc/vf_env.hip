// vf_env.hip -- the whole visual=False env step in one launch (gfx950).
//
// Replaces, per control step, the chain DroneGymEnvsBase.step -> DroneEnvsBase.step ->
// Dynamics.step -> update_collision -> get_success/get_reward -> done masks -> per-agent
// Python info loop -> examine()/reset_agent_by_id with its per-agent randomizer loop
// (envs/base/droneGymEnv.py:141-218,339-423; envs/base/droneEnv.py:237-288,345-379;
// envs/HoverEnv.py, envs/NavigationEnv.py, envs/RacingEnv.py).  One thread per agent; the
// env counters ride in the spare components of the dynamics granules, so the env step
// moves the same bytes as the bare dynamics step plus its outputs.
#include "vf_env_epilogue.hpp"

#pragma clang fp contract(off)

#ifdef VF_ENV_TRACE   // measurement build only (tools/exp_env_timeline.py): wall-clock (100 MHz) stamps of every main wave of k_env_step
__device__ unsigned long long* vf_env_trace_buf = nullptr;
__device__ unsigned vf_env_trace_cnt = 0, vf_env_trace_cap = 0;
extern "C" int vf_debug_env_trace(unsigned long long* buf, unsigned cap)
{
    unsigned zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(vf_env_trace_buf), &buf, sizeof(buf)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(vf_env_trace_cnt), &zero, sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(vf_env_trace_cap), &cap, sizeof(cap)) != hipSuccess) return -1;
    return 0;
}
#define VF_TR(i) tr[i] = __builtin_amdgcn_s_memrealtime()
#else
#define VF_TR(i)
#endif

namespace vf {

// LAZY_SLOT: the prefetched re-spawn copy is loaded only by the lanes that end an episode instead of by every lane
// (load_spawn_slot).  One wave per SIMD: an ending wave must not sit out the load latency -> every lane loads.  More than two waves
// per SIMD: the step is bandwidth-bound, a stalled wave costs nothing and 64 B per agent of loads nobody looks at cost 7 %
// (1 M agents: 99 vs 92.5 us) -> lazily.  Two kernels, not a branch: both epilogues in one kernel cost the small launch 0.1 us.
// The leading scalar arguments repeat fields of `g0` (and vf_dyn_cfg::delay_steps): this file is compiled with
// -mllvm -amdgpu-kernarg-preload-count=16, which has the dispatcher place the first 14 dwords of the kernel-argument segment in SGPRs
// before the wave starts (16 user SGPRs less the segment pointer; scalar arguments only -- a by-value struct is not preloaded).
// Everything the first burst of loads needs (slab, action rows, N, G, ring slot, drag granule) is among them, so the burst is issued without a single scalar-memory round trip;
// the rest of the kernel arguments and the two constant blocks arrive, in ONE batch, while it is in flight.  Before: kernel arguments
// (0.4 us), then the constant block for delay_steps (0.4 us), then the loads (timeline: profiles/r04_env_timeline.txt).
template <int KIND, int ACT, int INTEG, bool CTRL_DELAY, bool LAZY_SLOT = false>
__global__ __launch_bounds__(kBlock) void k_env_step(const vf_dyn_cfg* __restrict__ cp, const vf_env_cfg* __restrict__ ep, float* S,
                                                     const float4* action, int N, int G, int head, int helper, int delay_steps,
                                                     int g_drag, const EnvArgs g0)
{
    const vf_dyn_cfg& c = *cp;   // persistent device copies (vf_handles.hpp): L2-resident from launch to launch
    const vf_env_cfg& e = *ep;
    __shared__ __attribute__((aligned(16))) float tile[kBlock * 13];
    EnvArgs g = g0;
    g.d.S = S; g.d.action = action; g.d.N = N; g.d.G = G; g.d.head = head; g.helper = helper; g.d.g_drag = g_drag;
    // the blocks behind the g.helper main blocks: helper blocks (prefetched re-spawn), kHelperSpan agents per thread (a workgroup dispatch
    // costs more than their checks).  (Laying the helper's code and the episode-end blocks out behind the hot path with __builtin_expect --
    // the instruction cache is cold at every launch -- was measured in both regimes: no difference, profiles/r04_env_timeline.txt)
    if (g.helper != 0 && (int)blockIdx.x >= g.helper) {
        const int nbm = g.helper;
        const int base = ((int)blockIdx.x - nbm) * kHelperSpan * kBlock + (int)threadIdx.x;
#pragma unroll 1
        for (int k = 0; k < kHelperSpan; ++k) spawn_helper(e, g, base + k * kBlock);
        return;
    }
#ifdef VF_ENV_TRACE
    unsigned long long tr[13];
#endif
    VF_TR(0);
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < g.d.N;
    Agent s;
    Spares sp;
    float a[4], head_bits = 0.0f;
    ring_exchange_d(g.d, i, live, head_bits, a, nullptr, delay_steps);   // issued first: its two loads are the first values the controller needs
    load_agent<false>(g.d.S, g.d.G, i, s, sp);
    float4 dk0 = make_float4(0.f, 0.f, 0.f, 0.f), dk1 = dk0;
    if (g.d.g_drag >= 0) { dk0 = *granule(g.d.S, g.d.G, i, g.d.g_drag); dk1 = *granule(g.d.S, g.d.G, i, g.d.g_drag + 1); }
    // ... and behind the burst ONE batch of scalar loads: the remaining kernel-argument lines and the lines of the two constant blocks
    // (first touched one after the other on the way, each was a round trip a lone wave sits out in full)
#ifdef VF_ENV_TRACE
    VF_TR(1);                                        // vector loads issued
    prefetch_const_lines<(sizeof(vf_dyn_cfg) + 63) / 64, 2>(cp, ep, &ep->obs_mode);
    VF_TR(2);                                        // constant-block lines arrived
    prefetch_kernarg<sizeof(EnvArgs) + 56>();
    VF_TR(3);                                        // kernel-argument lines arrived
#else
    prefetch_kernarg_and_const_lines<sizeof(EnvArgs) + 56, (sizeof(vf_dyn_cfg) + 63) / 64, 2>(cp, ep, &ep->obs_mode);
#endif
    load_wind(c, g.d, i, live, s);
    if (delay_steps > 0) sp.vel = head_bits;
    float kl[3], kq[3];                                                       // drag_of
    if (g.d.g_drag >= 0) {
        kl[0] = dk0.y; kl[1] = dk0.z; kl[2] = dk0.w;
        kq[0] = dk1.y; kq[1] = dk1.z; kq[2] = dk1.w;
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { kl[k] = c.k_lin[k]; kq[k] = c.k_quad[k]; }
    }
#ifdef VF_ENV_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VF_TR(4);                                        // loads arrived
    struct TraceCk {
        unsigned long long* tr;
        __device__ __forceinline__ void head(int sub, const Agent& s) const
        {
            if (sub < 2) { asm volatile("" :: "v"(s.wm[0]), "v"(s.q.w) : "memory"); tr[5 + sub] = __builtin_amdgcn_s_memrealtime(); }
        }
        __device__ __forceinline__ void end(const Agent&) const {}
    };
    control_interval<ACT, INTEG, CTRL_DELAY>(c, s, a, kl, kq, g.d.vstrided != 0, TraceCk{tr});
    asm volatile("" :: "v"(s.p[0]), "v"(s.q.w), "v"(s.v[0]), "v"(s.w[0]) : "memory");
    VF_TR(7);                                        // interval done
#else
    control_interval<ACT, INTEG, CTRL_DELAY>(c, s, a, kl, kq, g.d.vstrided != 0);
#endif
    const int wave = threadIdx.x >> 6;
#ifdef VF_ENV_TRACE
    env_epilogue<KIND, true, 1, false, LAZY_SLOT>(c, e, g, i, live, s, sp, blockIdx.x * kBlock + wave * 64, tile + wave * 64 * 13, nullptr, nullptr, tr);
    asm volatile("" ::: "memory");
    VF_TR(11);                                       // stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VF_TR(12);                                       // stores acknowledged
    if ((threadIdx.x & 63) == 0 && vf_env_trace_buf) {
        const unsigned slot = atomicAdd(&vf_env_trace_cnt, 1u);
        if (slot < vf_env_trace_cap) {
            unsigned long long* o = vf_env_trace_buf + (size_t)slot * 16;
            for (int k = 0; k < 13; ++k) o[k] = tr[k];
            o[13] = ((unsigned long long)blockIdx.x << 8) | wave;
            unsigned xcc = 0;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned hwid = 0;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            o[14] = ((unsigned long long)xcc << 32) | hwid;
        }
    }
#else
    env_epilogue<KIND, true, 1, false, LAZY_SLOT>(c, e, g, i, live, s, sp, blockIdx.x * kBlock + wave * 64, tile + wave * 64 * 13);
#endif
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
    env_epilogue<KIND, true, 1, true>(c, e, g, i, live, s, sp, blockIdx.x * kBlock + wave * 64, tile + wave * 64 * 13);
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
__global__ __launch_bounds__(kBlock) void k_env_step_split(const vf_dyn_cfg* __restrict__ cp, const vf_env_cfg* __restrict__ ep, float* S,
                                                           const float4* action, int N, int G, int head, int helper, int delay_steps,
                                                           int g_drag, const EnvArgs g0)
{
    // (leading scalar arguments: preloaded into SGPRs, see k_env_step; `helper` unused here, kept for one launch signature)
    const vf_dyn_cfg& c = *cp;
    const vf_env_cfg& e = *ep;
    __shared__ __attribute__((aligned(16))) SplitShared shs[2];
    EnvArgs g = g0;
    g.d.S = S; g.d.action = action; g.d.N = N; g.d.G = G; g.d.head = head; g.d.g_drag = g_drag;
    const int grp = (threadIdx.x >> 6) & 1;
    SplitShared& sh = shs[grp];
    const int first = blockIdx.x * 128 + grp * 64;
    const int i = first + (threadIdx.x & 63);
    const bool live = i < g.d.N;
    if (threadIdx.x < 128) {
        split_rotation_wave<ACT, INTEG, CTRL_DELAY, sizeof(EnvArgs) + 56>(c, g.d, i, live, sh, delay_steps, ep, &ep->obs_mode);
        return;
    }
    Agent s;
    Spares sp;
    split_translation_wave<INTEG, sizeof(EnvArgs) + 56>(c, g.d, i, sh, s, sp, delay_steps, ep, &ep->obs_mode);
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
using EnvStepKernel = void (*)(const vf_dyn_cfg*, const vf_env_cfg*, float*, const float4*, int, int, int, int, int, int, const vf::EnvArgs);

template <int KIND, int ACT, bool LAZY>
EnvStepKernel pick_env_kernel_ka(const vf_dyn_cfg& c)
{
    const int key = (c.integrator == VF_INT_RK4 ? 2 : 0) | (c.ctrl_delay ? 1 : 0);
    switch (key) {
    case 0: return vf::k_env_step<KIND, ACT, VF_INT_EULER, false, LAZY>;
    case 1: return vf::k_env_step<KIND, ACT, VF_INT_EULER, true, LAZY>;
    case 2: return vf::k_env_step<KIND, ACT, VF_INT_RK4, false, LAZY>;
    default: return vf::k_env_step<KIND, ACT, VF_INT_RK4, true, LAZY>;
    }
}

template <int KIND, bool LAZY>
EnvStepKernel pick_env_kernel_kl(const vf_dyn_cfg& c)
{
    switch (c.action_type) {
    case VF_ACT_THRUST: return pick_env_kernel_ka<KIND, VF_ACT_THRUST, LAZY>(c);
    case VF_ACT_BODYRATE: return pick_env_kernel_ka<KIND, VF_ACT_BODYRATE, LAZY>(c);
    case VF_ACT_VELOCITY: return pick_env_kernel_ka<KIND, VF_ACT_VELOCITY, LAZY>(c);
    default: return pick_env_kernel_ka<KIND, VF_ACT_POSITION, LAZY>(c);
    }
}

template <int KIND>
EnvStepKernel pick_env_kernel_k(const vf_dyn_cfg& c, bool lazy)
{
    return lazy ? pick_env_kernel_kl<KIND, true>(c) : pick_env_kernel_kl<KIND, false>(c);
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
EnvStepKernel pick_env_split_k(const vf_dyn_cfg& c)
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

EnvStepKernel pick_env_split(const vf_env* h)
{
    switch (h->cfg.kind) {
    case VF_ENV_HOVER: return pick_env_split_k<VF_ENV_HOVER>(h->dyn.cfg);
    case VF_ENV_NAV: return pick_env_split_k<VF_ENV_NAV>(h->dyn.cfg);
    default: return pick_env_split_k<VF_ENV_RACING>(h->dyn.cfg);
    }
}


EnvStepKernel pick_env_kernel(const vf_env* h)
{
    const bool lazy = h->dyn.Npad > 2 * 65536;          // more than two waves per SIMD (k_env_step, LAZY_SLOT)
    switch (h->cfg.kind) {
    case VF_ENV_HOVER: return pick_env_kernel_k<VF_ENV_HOVER>(h->dyn.cfg, lazy);
    case VF_ENV_NAV: return pick_env_kernel_k<VF_ENV_NAV>(h->dyn.cfg, lazy);
    default: return pick_env_kernel_k<VF_ENV_RACING>(h->dyn.cfg, lazy);
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
        hipLaunchKernelGGL(pick_env_split(h), dim3(h->dyn.Npad / 128), dim3(vf::kBlock), 0, st, h->dyn.d_cfg, h->d_cfg, g.d.S, g.d.action, g.d.N, g.d.G,
                           g.d.head, 0, h->dyn.cfg.delay_steps, g.d.g_drag, g);
        h->stale_all = 1;
    } else {
        unsigned nb = h->dyn.Npad / vf::kBlock;
        if (h->g_spawn >= 0 && auto_reset) {     // prefetched re-spawn: main blocks read copy `par`, helper blocks refill the other
            const int par = (int)((h->dyn.tick + ahead) & 1);
            static const int mode = [] { const char* e = getenv("VISFLY_AMD_PREFETCH_MODE"); return e ? atoi(e) : 3; }();   // A/B: 1 = slot loads only, 2 = helper only
            if (mode & 1) g.g_spawn_rd = h->g_spawn + 4 * par;
            g.g_spawn_wr = h->g_spawn + 4 * (1 - par);
            if (mode & 2) { g.helper = (int)nb; nb += (nb + vf::kHelperSpan - 1) / vf::kHelperSpan; }
            // stale bits: (re)armed here when something re-spawned agents without maintaining them; not inside a stream capture
            // (the memset would replay with the graph) -- those launches run the helper's per-agent tag compare instead
            // OFF by default: they take the helper's reads from 16 to 3 B per agent-step but cost the reset regime 0.3 us per launch (two
            // atomics in every ending wave, a dependent load more in front of the helper's refills): profiles/r04_env_quad.txt
            static const bool bits_off = [] { const char* e = getenv("VISFLY_AMD_STALE_BITS"); return !(e && atoi(e) == 1); }();
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            const bool bits_wanted = h->d_stale && !bits_off && (mode & 2);
            if (bits_wanted) (void)hipStreamIsCapturing(st, &cap);        // (a runtime call per launch: only when the bits are on)
            if (bits_wanted && cap == hipStreamCaptureStatusNone) {
                const int tiles = h->dyn.Npad / 64;
                if (h->stale_all) {
                    VF_HIP(hipMemsetAsync(h->d_stale, 0xFF, (size_t)2 * tiles * sizeof(unsigned long long), st));
                    h->stale_all = 0;
                }
                g.stale = h->d_stale;
                g.n_tiles = tiles;
                g.stale_wr = 1 - par;
            } else {
                h->stale_all = 1;        // these launches re-spawn agents behind the bits' back
            }
        } else {
            h->stale_all = 1;
        }
        hipLaunchKernelGGL(pick_env_kernel(h), dim3(nb), dim3(vf::kBlock), 0, st, h->dyn.d_cfg, h->d_cfg, g.d.S, g.d.action, g.d.N, g.d.G,
                           g.d.head, g.helper, h->dyn.cfg.delay_steps, g.d.g_drag, g);
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
    if (rc == VF_OK && h->g_spawn >= 0 &&
        hipMalloc(reinterpret_cast<void**>(&h->d_stale), (size_t)2 * (h->dyn.Npad / 64) * sizeof(unsigned long long)) != hipSuccess) {
        h->d_stale = nullptr;            // no bits: the helper compares every tag, as before
        (void)hipGetLastError();
    }
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
    if (h->d_stale) (void)hipFree(h->d_stale);
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
    h->stale_all = 1;            // episode counters / spawn copies change behind the stale bits' back
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
    h->stale_all = 1;        // re-spawns inside the fused launch are not reflected in the stale bits
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
    h->stale_all = 1;        // re-spawns of the split step are not reflected in the stale bits
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
