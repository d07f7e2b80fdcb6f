// vf_dyn.hip -- fused quadrotor control-interval kernel + reset kernel + C-ABI (gfx950).
//
// Replaces Dynamics.step / Dynamics.reset (envs/base/dynamics.py:218-269,319-382) and the
// Quaternion / Integrator helpers it drives (utils/maths.py).  The reference issues ~3.8k
// aten ops per control step; here one launch does the whole interval: one thread per
// agent, state in VGPRs for all sub-steps, the slab tiled per wavefront (one contiguous
// chunk of 16-byte granules per wave, include/visfly_amd.h), the AoS (N,13) observation
// staged through LDS for coalesced stores.
#include "vf_common.hpp"
#include "vf_dyn_device.hpp"
#include "vf_handles.hpp"

#pragma clang fp contract(off)

namespace vf {

char* err_buf()
{
    static thread_local char buf[512] = "";
    return buf;
}

// grid covers the padded agent count (multiple of 64); pad lanes integrate an inert hover state
template <int ACT, int INTEG, bool CTRL_DELAY>
__global__ __launch_bounds__(kBlock) void k_dyn_step(const vf_dyn_cfg* __restrict__ cp, float* S, const float4* action, float* obs, int N, int G,
                                                     int g_drag, int head, int delay_steps, const DynArgs g0)
{
    // leading scalar arguments = fields of g0 + vf_dyn_cfg::delay_steps, preloaded into SGPRs by the dispatcher (this file is compiled with
    // -mllvm -amdgpu-kernarg-preload-count=16): the first burst of loads is issued without a scalar-memory round trip, the remaining
    // kernel-argument lines and the constant block arrive in one batch behind it (k_env_step, vf_env.hip)
    const vf_dyn_cfg& c = *cp;   // persistent device copy (vf_handles.hpp): stays L2-resident from launch to launch
    __shared__ __attribute__((aligned(16))) float tile[kBlock * 13];
    DynArgs g = g0;
    g.S = S; g.action = action; g.obs = obs; g.N = N; g.G = G; g.g_drag = g_drag; g.head = head;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < g.N;
    Agent s;
    Spares sp;
    float a[4], head_bits = 0.0f;
    ring_exchange_d(g, i, live, head_bits, a, nullptr, delay_steps);   // issued first: its two loads are the first values the controller needs
    load_agent<false>(g.S, g.G, i, s, sp);
    float4 dk0 = make_float4(0.f, 0.f, 0.f, 0.f), dk1 = dk0;
    if (g.g_drag >= 0) { dk0 = *granule(g.S, g.G, i, g.g_drag); dk1 = *granule(g.S, g.G, i, g.g_drag + 1); }
    prefetch_kernarg_and_const_lines<sizeof(DynArgs) + 56, (sizeof(vf_dyn_cfg) + 63) / 64, 0>(cp, cp, cp);
    load_wind(c, g, i, live, s);
    if (delay_steps > 0) sp.vel = head_bits;
    float kl[3], kq[3];                                                // drag_of
    if (g.g_drag >= 0) {
        kl[0] = dk0.y; kl[1] = dk0.z; kl[2] = dk0.w;
        kq[0] = dk1.y; kq[1] = dk1.z; kq[2] = dk1.w;
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { kl[k] = c.k_lin[k]; kq[k] = c.k_quad[k]; }
    }
    control_interval<ACT, INTEG, CTRL_DELAY>(c, s, a, kl, kq, g.vstrided != 0);
    store_agent(g.S, g.G, i, s, sp);
    if (g.obs) {
        float o[13];
        obs_row(c, s, o);
        const int wave = threadIdx.x >> 6;
        store_rows_coalesced<13>(g.obs, g.N, blockIdx.x * kBlock + wave * 64, o, tile + wave * 64 * 13);
    }
}

// Two-wave variant (see SplitShared): 256-thread workgroups = 2 rotation + 2 translation waves for 128
// agents, so that every workgroup puts exactly one wave on each SIMD of its CU.
template <int ACT, int INTEG, bool CTRL_DELAY>
__global__ __launch_bounds__(kBlock) void k_dyn_step_split(const vf_dyn_cfg* __restrict__ cp, float* S, const float4* action, float* obs, int N, int G,
                                                           int g_drag, int head, int delay_steps, const DynArgs g0)
{
    const vf_dyn_cfg& c = *cp;   // (leading scalar arguments: preloaded into SGPRs, see k_dyn_step)
    __shared__ __attribute__((aligned(16))) SplitShared shs[2];
    DynArgs g = g0;
    g.S = S; g.action = action; g.obs = obs; g.N = N; g.G = G; g.g_drag = g_drag; g.head = head;
    const int grp = (threadIdx.x >> 6) & 1;
    SplitShared& sh = shs[grp];
    const int first = blockIdx.x * 128 + grp * 64;
    const int i = first + (threadIdx.x & 63);
    const bool live = i < g.N;
    if (threadIdx.x < 128) {
        split_rotation_wave<ACT, INTEG, CTRL_DELAY>(c, g, i, live, sh, delay_steps);
        return;
    }
    Agent s;
    Spares sp;
    split_translation_wave<INTEG>(c, g, i, sh, s, sp, delay_steps);
    store_agent(g.S, g.G, i, s, sp);
    if (g.obs) {
        float o[13];
        obs_row(c, s, o);
        store_rows_coalesced<13>(g.obs, g.N, first, o, sh.tile);
    }
}

struct ResetArgs {
    int N, Npad, k, G, g_drag;
    float* S;
    const int* idx;
    const float *pos, *quat, *vel, *omg, *mot, *thr, *t, *t_rand, *klin, *kquad;
};

// One thread per reset entry; a full reset (idx == null) also initialises the pad lanes.
__global__ __launch_bounds__(kBlock) void k_dyn_reset(const vf_dyn_cfg c, const ResetArgs r)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= r.k) return;
    const int i = r.idx ? r.idx[j] : j;
    const bool given = j < r.N || r.idx;  // pad lanes of a full reset take defaults
    const size_t j3 = 3 * (size_t)j, j4 = 4 * (size_t)j;
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = make_float4(1.f, 0.f, 0.f, 0.f);
    float4 g2 = *granule(r.S, r.G, i, VF_G_VEL), g3 = *granule(r.S, r.G, i, VF_G_OMG);
    float4 g6 = *granule(r.S, r.G, i, VF_G_AACC), g7 = *granule(r.S, r.G, i, VF_G_ACC);
    g2.y = g2.z = g2.w = 0.f;
    if (!r.idx) g2.x = 0.f;  // ring head
    g3.y = g3.z = g3.w = 0.f;
    g6.y = g6.z = g6.w = 0.f;  // dynamics.py:239-240,258-259
    g7.y = g7.z = g7.w = 0.f;
    float4 g4 = make_float4(c.w_init, c.w_init, c.w_init, c.w_init);
    float4 g5 = make_float4(c.T_init, c.T_init, c.T_init, c.T_init);
    if (given) {
        if (r.pos) { g0.y = r.pos[j3]; g0.z = r.pos[j3 + 1]; g0.w = r.pos[j3 + 2]; }
        if (r.quat) g1 = make_float4(r.quat[j4], r.quat[j4 + 1], r.quat[j4 + 2], r.quat[j4 + 3]);
        if (r.vel) { g2.y = r.vel[j3]; g2.z = r.vel[j3 + 1]; g2.w = r.vel[j3 + 2]; }
        if (r.omg) { g3.y = r.omg[j3]; g3.z = r.omg[j3 + 1]; g3.w = r.omg[j3 + 2]; }
        if (r.mot) g4 = make_float4(r.mot[j4], r.mot[j4 + 1], r.mot[j4 + 2], r.mot[j4 + 3]);
        if (r.thr) g5 = make_float4(r.thr[j4], r.thr[j4 + 1], r.thr[j4 + 2], r.thr[j4 + 3]);
        if (r.t) g0.x = r.t[j];
        else if (r.idx && r.t_rand) g0.x = 0.0f + r.t_rand[j] * 3.14f * 2.0f;  // dynamics.py:256
    }
    *granule(r.S, r.G, i, VF_G_POS) = g0;
    *granule(r.S, r.G, i, VF_G_QUAT) = g1;
    *granule(r.S, r.G, i, VF_G_VEL) = g2;
    *granule(r.S, r.G, i, VF_G_OMG) = g3;
    *granule(r.S, r.G, i, VF_G_MOT) = g4;
    *granule(r.S, r.G, i, VF_G_THR) = g5;
    *granule(r.S, r.G, i, VF_G_AACC) = g6;
    *granule(r.S, r.G, i, VF_G_ACC) = g7;
    for (int s = 0; s < c.delay_steps; ++s)  // :243,262-263
        *granule(r.S, r.G, i, VF_G_RING + s) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r.g_drag >= 0) {
        if (given && r.klin) {
            *granule(r.S, r.G, i, r.g_drag) = make_float4(0.f, r.klin[j3], r.klin[j3 + 1], r.klin[j3 + 2]);
            *granule(r.S, r.G, i, r.g_drag + 1) = make_float4(0.f, r.kquad[j3], r.kquad[j3 + 1], r.kquad[j3 + 2]);
        } else if (!r.idx) {
            *granule(r.S, r.G, i, r.g_drag) = make_float4(0.f, c.k_lin[0], c.k_lin[1], c.k_lin[2]);
            *granule(r.S, r.G, i, r.g_drag + 1) = make_float4(0.f, c.k_quad[0], c.k_quad[1], c.k_quad[2]);
        }
    }
}

}  // namespace vf


namespace {

using StepKernel = void (*)(const vf_dyn_cfg*, const vf::DynArgs);
using StepKernel1 = void (*)(const vf_dyn_cfg*, float*, const float4*, float*, int, int, int, int, int, const vf::DynArgs);

StepKernel1 pick_split_kernel(const vf_dyn_cfg& c)
{
    const int key = (c.action_type == VF_ACT_BODYRATE ? 4 : 0) | (c.integrator == VF_INT_RK4 ? 2 : 0) |
                    (c.ctrl_delay ? 1 : 0);
    switch (key) {
    case 0: return vf::k_dyn_step_split<VF_ACT_THRUST, VF_INT_EULER, false>;
    case 1: return vf::k_dyn_step_split<VF_ACT_THRUST, VF_INT_EULER, true>;
    case 2: return vf::k_dyn_step_split<VF_ACT_THRUST, VF_INT_RK4, false>;
    case 3: return vf::k_dyn_step_split<VF_ACT_THRUST, VF_INT_RK4, true>;
    case 4: return vf::k_dyn_step_split<VF_ACT_BODYRATE, VF_INT_EULER, false>;
    case 5: return vf::k_dyn_step_split<VF_ACT_BODYRATE, VF_INT_EULER, true>;
    case 6: return vf::k_dyn_step_split<VF_ACT_BODYRATE, VF_INT_RK4, false>;
    default: return vf::k_dyn_step_split<VF_ACT_BODYRATE, VF_INT_RK4, true>;
    }
}

template <int ACT>
StepKernel1 pick_step_kernel_a(const vf_dyn_cfg& c)
{
    const int key = (c.integrator == VF_INT_RK4 ? 2 : 0) | (c.ctrl_delay ? 1 : 0);
    switch (key) {
    case 0: return vf::k_dyn_step<ACT, VF_INT_EULER, false>;
    case 1: return vf::k_dyn_step<ACT, VF_INT_EULER, true>;
    case 2: return vf::k_dyn_step<ACT, VF_INT_RK4, false>;
    default: return vf::k_dyn_step<ACT, VF_INT_RK4, true>;
    }
}

StepKernel1 pick_step_kernel(const vf_dyn_cfg& c)
{
    switch (c.action_type) {
    case VF_ACT_THRUST: return pick_step_kernel_a<VF_ACT_THRUST>(c);
    case VF_ACT_BODYRATE: return pick_step_kernel_a<VF_ACT_BODYRATE>(c);
    case VF_ACT_VELOCITY: return pick_step_kernel_a<VF_ACT_VELOCITY>(c);
    default: return pick_step_kernel_a<VF_ACT_POSITION>(c);
    }
}

int launch_step(vf_dyn* h, const float* action, float* state_out, hipStream_t st)
{
    vf::DynArgs g{h->N, h->G, h->g_drag, h->S, reinterpret_cast<const float4*>(action), state_out, vf::ring_head(h),
                  reinterpret_cast<const float4*>(h->wind), h->vel_strided};
    h->tick += 1;
    if (vf::use_split(h->Npad, h->cfg))
        hipLaunchKernelGGL(pick_split_kernel(h->cfg), dim3(h->Npad / 128), dim3(vf::kBlock), 0, st, h->d_cfg, g.S, g.action, g.obs, g.N, g.G,
                           g.g_drag, g.head, h->cfg.delay_steps, g);
    else
        hipLaunchKernelGGL(pick_step_kernel(h->cfg), dim3(h->Npad / vf::kBlock), dim3(vf::kBlock), 0, st, h->d_cfg, g.S, g.action, g.obs, g.N, g.G,
                           g.g_drag, g.head, h->cfg.delay_steps, g);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

}  // namespace

extern "C" {

const char* vf_last_error(void) { return vf::err_buf(); }

int32_t vf_abi_version(void) { return VF_ABI_VERSION; }

int vf_dyn_create(const vf_dyn_cfg* cfg, int32_t N, int32_t per_agent_drag, vf_dyn** out)
{
    if (!cfg || !out || N <= 0) return vf::fail(VF_EINVAL, "vf_dyn_create: null argument or N <= 0");
    if (int rc = vf::check_dyn_cfg(cfg)) return rc;
    vf_dyn* h = new vf_dyn;
    vf::init_dyn_handle(h, cfg, N, per_agent_drag, 0);
    if (int rc = vf::upload_cfg(h->cfg, &h->d_cfg)) {
        delete h;
        return rc;
    }
    *out = h;
    return VF_OK;
}

void vf_dyn_destroy(vf_dyn* h)
{
    if (!h) return;
    vf::release_cfg(&h->d_cfg);
    vf::release_cfg(&h->d_env_dummy);
    delete h;
}

int32_t vf_dyn_granules(const vf_dyn* h) { return h ? h->G : 0; }

int64_t vf_dyn_slab_floats(const vf_dyn* h) { return h ? (int64_t)h->Npad * h->G * 4 : 0; }

int vf_dyn_bind(vf_dyn* h, float* slab)
{
    if (!h || !slab) return vf::fail(VF_EINVAL, "vf_dyn_bind: null handle or slab");
    if (reinterpret_cast<uintptr_t>(slab) % 16) return vf::fail(VF_EINVAL, "vf_dyn_bind: slab must be 16-byte aligned");
    h->S = slab;
    return VF_OK;
}

int vf_dyn_step(vf_dyn* h, const float* action, float* state_out, vf_stream_t stream)
{
    if (!h || !action) return vf::fail(VF_EINVAL, "vf_dyn_step: null handle or action");
    if (!h->S) return vf::fail(VF_ESTATE, "vf_dyn_step: vf_dyn_bind has not been called");
    if (int rc = vf::refuse_capture(h, vf::as_stream(stream), "vf_dyn_step")) return rc;
    return launch_step(h, action, state_out, vf::as_stream(stream));
}

int32_t vf_dyn_ring_phase(const vf_dyn* h) { return h ? vf::ring_head(h) : 0; }

int vf_dyn_set_ring_phase(vf_dyn* h, int32_t phase)
{
    if (!h) return vf::fail(VF_EINVAL, "vf_dyn_set_ring_phase: null handle");
    return vf::set_ring_phase(h, phase, "vf_dyn_set_ring_phase");
}

int vf_dyn_reset(vf_dyn* h, const int32_t* idx, int32_t k, const float* pos, const float* quat, const float* vel,
                 const float* omg, const float* mot, const float* thr, const float* t, const float* t_rand,
                 const float* klin, const float* kquad, vf_stream_t stream)
{
    if (!h) return vf::fail(VF_EINVAL, "vf_dyn_reset: null handle");
    if (!h->S) return vf::fail(VF_ESTATE, "vf_dyn_reset: vf_dyn_bind has not been called");
    if ((klin == nullptr) != (kquad == nullptr)) return vf::fail(VF_EINVAL, "vf_dyn_reset: klin and kquad go together");
    if (klin && h->g_drag < 0) return vf::fail(VF_EINVAL, "vf_dyn_reset: handle was created without per-agent drag");
    const int n = idx ? k : h->Npad;
    if (n < 0) return vf::fail(VF_EINVAL, "vf_dyn_reset: k < 0");
    hipStream_t st = vf::as_stream(stream);
    if (n == 0) return VF_OK;
    if (!idx) h->tick = 0;   // full reset: every head word goes to 0 (k_dyn_reset), and so does the launch-uniform phase
    if (!idx) h->vel_strided = vel ? 1 : 0;   // reset(vel=...) stores the strided view vel.T, reset() contiguous zeros (dynamics.py:236)
    vf::ResetArgs r{h->N, h->Npad, n, h->G, h->g_drag, h->S, idx, pos, quat, vel, omg, mot, thr, t, t_rand, klin, kquad};
    hipLaunchKernelGGL(vf::k_dyn_reset, dim3(vf::blocks_for(n)), dim3(vf::kBlock), 0, st, h->cfg, r);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_dyn_set_wind(vf_dyn* h, const float* wind_Nx4)
{
    if (!h) return vf::fail(VF_EINVAL, "vf_dyn_set_wind: null handle");
    if (reinterpret_cast<uintptr_t>(wind_Nx4) & 15) return vf::fail(VF_EINVAL, "vf_dyn_set_wind: rows must be 16-byte aligned");
    h->wind = wind_Nx4;
    return VF_OK;
}

int vf_dyn_time_steps(vf_dyn* h, const float* action, float* state_out, int32_t iters, vf_stream_t stream,
                      float* mean_us)
{
    if (!h || !action || !mean_us || iters <= 0) return vf::fail(VF_EINVAL, "vf_dyn_time_steps: bad argument");
    if (!h->S) return vf::fail(VF_ESTATE, "vf_dyn_time_steps: vf_dyn_bind has not been called");
    hipStream_t st = vf::as_stream(stream);
    hipEvent_t e0, e1;
    VF_HIP(hipEventCreate(&e0));
    VF_HIP(hipEventCreate(&e1));
    VF_HIP(hipEventRecord(e0, st));
    for (int it = 0; it < iters; ++it) {
        int rc = launch_step(h, action, state_out, st);
        if (rc != VF_OK) return rc;
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
