// Host-side handle layouts shared by the translation units of libvisfly_amd.so.
#pragma once
#include "vf_common.hpp"

struct vf_dyn {
    vf_dyn_cfg cfg;
    int N, Npad, G, g_drag, g_extra;  // g_extra: first granule after the dynamics ones (env layer)
    float* S = nullptr;
    // control steps since the last full reset.  Every agent's delay-ring head equals tick % delay_steps (all agents push
    // once per step; an indexed reset zeroes the agent's slots, after which the head position is immaterial), so the step
    // kernels take the slot index from the launch arguments -- the ring-slot load no longer waits for the velocity granule
    // that carries the per-agent copy (still written: the adjoint kernel reads it from its tape).
    long long tick = 0;
    // Layout of the reference's `_velocity` tensor, which decides WHICH atan2 torch runs for the velocity action type's auto-yaw
    // (dynamics.py:423-427): Dynamics.reset stores `vel.T` -- a strided view -- when velocities are passed (:236; every env
    // reset does) and contiguous zeros otherwise; in-place updates and clamp() keep that layout until the next full reset.
    // Strided operands take torch's scalar loop = glibc's atan2f, contiguous ones the SLEEF loop (vf_xmath.hpp).
    int vel_strided = 0;
    // device copy of cfg (vf_dyn_create / vf_env_create): the step kernels read their ~600 B of constants through this pointer
    // instead of by-value kernel arguments.  A by-value block lives at a fresh kernarg address every launch, so each wave's
    // scalar loads miss all the way to HBM; the persistent copy stays in the XCD L2s from launch to launch (65 536 agents:
    // 11.5 -> 10.7 us per step, tools/env_step_probe.hip; a lone 64-agent wave pays ~1 us for the extra dependent load).
    vf_dyn_cfg* d_cfg = nullptr;
    struct vf_env_cfg* d_env_dummy = nullptr;   // vf_dyn_step_bwd only: the adjoint kernel's (unused) env constant block
    // per-agent wind rows (N x 4 floats, caller-owned device memory; vf_dyn_set_wind / vf_env_set_wind) or null = cfg.wind
    const float* wind = nullptr;
};

struct vf_env {
    vf_dyn dyn;
    vf_env_cfg cfg;
    int g_race;  // racing granule or -1
    int g_spawn = -1;  // first of the 2 x 4 prefetched re-spawn granules or -1 (vf_env_cfg.spawn_prefetch)
    // prefetched re-spawn, r04: one "may be stale" bit per (spawn copy, agent), a 64-bit word per wave tile -- [2][Npad / 64], owned by the
    // handle.  The helper blocks of a step launch read ONE word per 64 agents and look at an agent's episode counter / copy tag only
    // where its bit is set (r03: two words per agent per launch, 32 B of line traffic per agent-step); an ending agent sets its bit
    // in both copies.  A hint only: the tag compare at consumption decides validity.  stale_all: every bit must be set before the
    // next helper pass (after a reset, or after launches that re-spawn agents without maintaining the bits)
    unsigned long long* d_stale = nullptr;
    int stale_all = 1;
    vf_env_cfg* d_cfg = nullptr;   // device copy of cfg, see vf_dyn::d_cfg
};

namespace vf {

inline int check_dyn_cfg(const vf_dyn_cfg* cfg)
{
    if (cfg->action_type < VF_ACT_THRUST || cfg->action_type > VF_ACT_POSITION)
        return fail(VF_EINVAL, "action_type %d not supported (thrust=0, bodyrate=1, velocity=2, position=3)",
                    cfg->action_type);
    if (cfg->integrator != VF_INT_EULER && cfg->integrator != VF_INT_RK4)
        return fail(VF_EINVAL, "integrator %d not supported (euler=0, rk4=1)", cfg->integrator);
    if (cfg->interval_steps <= 0 || cfg->delay_steps < 0 || cfg->delay_steps > 64)
        return fail(VF_EINVAL, "bad interval_steps/delay_steps");
    return VF_OK;
}

// granule budget: 8 dynamics + delay ring + optional per-agent drag pair + `extra` env granules
inline void init_dyn_handle(vf_dyn* h, const vf_dyn_cfg* cfg, int N, int per_agent_drag, int extra)
{
    h->cfg = *cfg;
    h->N = N;
    // pad to whole workgroups so that every lane of every wave owns a (possibly inert) agent
    h->Npad = (N + kBlock - 1) / kBlock * kBlock;
    h->g_drag = per_agent_drag ? VF_G_FIXED + cfg->delay_steps : -1;
    h->g_extra = VF_G_FIXED + cfg->delay_steps + (per_agent_drag ? 2 : 0);
    h->G = h->g_extra + extra;
    h->S = nullptr;
    h->tick = 0;
    h->vel_strided = 0;
}

// device copies of the constant blocks (current HIP device); freed by release_cfg
template <class T>
inline int upload_cfg(const T& host, T** dev)
{
    VF_HIP(hipMalloc(reinterpret_cast<void**>(dev), sizeof(T)));
    VF_HIP(hipMemcpy(*dev, &host, sizeof(T), hipMemcpyHostToDevice));
    return VF_OK;
}
template <class T>
inline void release_cfg(T** dev)
{
    if (*dev) (void)hipFree(*dev);
    *dev = nullptr;
}

// ring slot of the step that is `ahead` launches after the next one
inline int ring_head(const vf_dyn* h, int ahead = 0)
{
    return h->cfg.delay_steps > 0 ? (int)((h->tick + ahead) % h->cfg.delay_steps) : 0;
}

// vf_dyn_step / vf_env_step inside somebody else's stream capture: the launch would bake one ring slot into the graph
inline int refuse_capture(const vf_dyn* h, hipStream_t st, const char* who)
{
    if (h->cfg.delay_steps <= 1) return VF_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
        (void)hipGetLastError();
        return VF_OK;
    }
    if (cs != hipStreamCaptureStatusNone)
        return fail(VF_ESTATE, "%s: the stream is capturing and delay_steps = %d: a captured launch bakes ONE delay-ring slot "
                               "into the graph (every replay would reuse it).  Use vf_env_graph_create / vf_env_graph_launch",
                    who, h->cfg.delay_steps);
    return VF_OK;
}

inline int set_ring_phase(vf_dyn* h, int phase, const char* who)
{
    const int D = h->cfg.delay_steps;
    if (phase < 0 || (D > 0 ? phase >= D : phase != 0)) return fail(VF_EINVAL, "%s: phase %d outside [0, %d)", who, phase, D > 0 ? D : 1);
    h->tick = phase;
    return VF_OK;
}

}  // namespace vf
