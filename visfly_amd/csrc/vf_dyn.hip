// vf_dyn.hip -- fused quadrotor control-interval kernel + reset kernel + C-ABI (gfx950).
//
// Replaces Dynamics.step / Dynamics.reset (envs/base/dynamics.py:218-269,319-382) and the
// Quaternion / Integrator helpers it drives (utils/maths.py).  The reference issues ~3.8k
// aten ops per control step; here one launch does the whole interval: one thread per
// agent, state in VGPRs for all sub-steps, coalesced SoA row loads/stores, the AoS
// (N,13) observation staged through LDS for coalesced stores.
#include "vf_common.hpp"
#include "vf_dyn_device.hpp"

#pragma clang fp contract(off)

namespace vf {

char* err_buf()
{
    static thread_local char buf[512] = "";
    return buf;
}

struct DynArgs {
    int N;
    float* S;           // [VF_ROWS][N]
    float* Q;           // [D][4][N] ring
    int* ctl;           // ctl[0] = ring head, ctl[1] = arrived blocks
    const float* klin;  // [3][N] or null
    const float* kquad;
    const float4* action;  // (N,4)
    float* obs;            // (N,13) or null
};

// Pops the oldest action of agent i from the ring slot and pushes the new one
// (dynamics.py:323-328).  The ring head is a device word advanced by the last block to
// finish, so the launch is replayable from a hipGraph without host-side arguments.
__device__ __forceinline__ void ring_exchange(const vf_dyn_cfg& c, const DynArgs& g, int i, float* a)
{
    const float4 an = g.action[i];
    if (c.delay_steps > 0) {
        const int slot = g.ctl[0];
        float* qs = g.Q + ((size_t)slot * 4) * g.N + i;
        a[0] = qs[0];
        a[1] = qs[(size_t)g.N];
        a[2] = qs[(size_t)2 * g.N];
        a[3] = qs[(size_t)3 * g.N];
        qs[0] = an.x;
        qs[(size_t)g.N] = an.y;
        qs[(size_t)2 * g.N] = an.z;
        qs[(size_t)3 * g.N] = an.w;
    } else {
        a[0] = an.x; a[1] = an.y; a[2] = an.z; a[3] = an.w;
    }
}

__device__ __forceinline__ void ring_advance(const vf_dyn_cfg& c, int* ctl)
{
    // every wave of this block has read ctl[0] (callers place a __syncthreads() before)
    if (c.delay_steps > 0 && threadIdx.x == 0) {
        const int head = ctl[0];
        const unsigned prev = atomicAdd(reinterpret_cast<unsigned*>(ctl + 1), 1u);
        if (prev == gridDim.x - 1) {
            ctl[1] = 0;
            ctl[0] = head + 1 == c.delay_steps ? 0 : head + 1;
        }
    }
}

template <int ACT, int INTEG, bool CTRL_DELAY>
__global__ __launch_bounds__(kBlock) void k_dyn_step(const vf_dyn_cfg c, const DynArgs g)
{
    __shared__ float tile[kBlock * 13];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < g.N;
    float o[13];
    if (live) {
        float a[4];
        ring_exchange(c, g, i, a);
        Agent s;
        load_agent(g.S, g.N, i, s);
        float kl[3], kq[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            kl[k] = g.klin ? g.klin[(size_t)k * g.N + i] : c.k_lin[k];
            kq[k] = g.kquad ? g.kquad[(size_t)k * g.N + i] : c.k_quad[k];
        }
        control_interval<ACT, INTEG, CTRL_DELAY>(c, s, a, kl, kq);
        store_agent(g.S, g.N, i, s);
        obs_row(c, s, o);
    }
    if (g.obs) store_rows_coalesced<13>(g.obs, g.N, blockIdx.x * kBlock, o, tile);
    else __syncthreads();
    ring_advance(c, g.ctl);
}

struct ResetArgs {
    int N, k;
    float* S;
    float* Q;
    const int* idx;
    const float *pos, *quat, *vel, *omg, *mot, *thr, *t, *t_rand;
};

__global__ __launch_bounds__(kBlock) void k_dyn_reset(const vf_dyn_cfg c, const ResetArgs r)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= r.k) return;
    const int i = r.idx ? r.idx[j] : j;
    float* b = r.S + i;
    const size_t N = r.N;
#pragma unroll
    for (int d = 0; d < 3; ++d) b[(VF_POS + d) * N] = r.pos ? r.pos[3 * (size_t)j + d] : 0.0f;
#pragma unroll
    for (int d = 0; d < 4; ++d) b[(VF_QUAT + d) * N] = r.quat ? r.quat[4 * (size_t)j + d] : (d == 0 ? 1.0f : 0.0f);
#pragma unroll
    for (int d = 0; d < 3; ++d) b[(VF_VEL + d) * N] = r.vel ? r.vel[3 * (size_t)j + d] : 0.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) b[(VF_OMG + d) * N] = r.omg ? r.omg[3 * (size_t)j + d] : 0.0f;
#pragma unroll
    for (int d = 0; d < 4; ++d) b[(VF_MOT + d) * N] = r.mot ? r.mot[4 * (size_t)j + d] : c.w_init;
#pragma unroll
    for (int d = 0; d < 4; ++d) b[(VF_THR + d) * N] = r.thr ? r.thr[4 * (size_t)j + d] : c.T_init;
#pragma unroll
    for (int d = 0; d < 3; ++d) b[(VF_AACC + d) * N] = 0.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) b[(VF_ACC + d) * N] = 0.0f;
    float t = 0.0f;
    if (r.t) t = r.t[j];
    else if (r.idx && r.t_rand) t = 0.0f + r.t_rand[j] * 3.14f * 2.0f;  // dynamics.py:256
    b[VF_T * N] = t;
    if (r.Q)
        for (int s = 0; s < c.delay_steps * 4; ++s) r.Q[(size_t)s * N + i] = 0.0f;  // :243,262-263
}

}  // namespace vf

struct vf_dyn {
    vf_dyn_cfg cfg;
    int N;
    float* S = nullptr;
    float* Q = nullptr;
    const float* klin = nullptr;
    const float* kquad = nullptr;
    int* ctl = nullptr;  // device: {ring head, arrival counter}
};

namespace {

using StepKernel = void (*)(const vf_dyn_cfg, const vf::DynArgs);

StepKernel pick_step_kernel(const vf_dyn_cfg& c)
{
    const int key = (c.action_type == VF_ACT_BODYRATE ? 4 : 0) | (c.integrator == VF_INT_RK4 ? 2 : 0) |
                    (c.ctrl_delay ? 1 : 0);
    switch (key) {
    case 0: return vf::k_dyn_step<VF_ACT_THRUST, VF_INT_EULER, false>;
    case 1: return vf::k_dyn_step<VF_ACT_THRUST, VF_INT_EULER, true>;
    case 2: return vf::k_dyn_step<VF_ACT_THRUST, VF_INT_RK4, false>;
    case 3: return vf::k_dyn_step<VF_ACT_THRUST, VF_INT_RK4, true>;
    case 4: return vf::k_dyn_step<VF_ACT_BODYRATE, VF_INT_EULER, false>;
    case 5: return vf::k_dyn_step<VF_ACT_BODYRATE, VF_INT_EULER, true>;
    case 6: return vf::k_dyn_step<VF_ACT_BODYRATE, VF_INT_RK4, false>;
    default: return vf::k_dyn_step<VF_ACT_BODYRATE, VF_INT_RK4, true>;
    }
}

int launch_step(vf_dyn* h, const float* action, float* state_out, hipStream_t st)
{
    vf::DynArgs g{h->N, h->S, h->Q, h->ctl, h->klin, h->kquad, reinterpret_cast<const float4*>(action), state_out};
    hipLaunchKernelGGL(pick_step_kernel(h->cfg), dim3(vf::blocks_for(h->N)), dim3(vf::kBlock), 0, st, h->cfg, g);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

}  // namespace

extern "C" {

const char* vf_last_error(void) { return vf::err_buf(); }

int32_t vf_abi_version(void) { return VF_ABI_VERSION; }

int vf_dyn_create(const vf_dyn_cfg* cfg, int32_t N, vf_dyn** out)
{
    if (!cfg || !out || N <= 0) return vf::fail(VF_EINVAL, "vf_dyn_create: null argument or N <= 0");
    if (cfg->action_type != VF_ACT_THRUST && cfg->action_type != VF_ACT_BODYRATE)
        return vf::fail(VF_EINVAL, "vf_dyn_create: action_type %d not supported (thrust=0, bodyrate=1)", cfg->action_type);
    if (cfg->integrator != VF_INT_EULER && cfg->integrator != VF_INT_RK4)
        return vf::fail(VF_EINVAL, "vf_dyn_create: integrator %d not supported (euler=0, rk4=1)", cfg->integrator);
    if (cfg->interval_steps <= 0 || cfg->delay_steps < 0)
        return vf::fail(VF_EINVAL, "vf_dyn_create: bad interval_steps/delay_steps");
    vf_dyn* h = new vf_dyn;
    h->cfg = *cfg;
    h->N = N;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&h->ctl), 2 * sizeof(int));
    if (e == hipSuccess) e = hipMemset(h->ctl, 0, 2 * sizeof(int));
    if (e != hipSuccess) {
        delete h;
        return vf::fail(VF_EHIP, "vf_dyn_create: hipMalloc/hipMemset failed: %s", hipGetErrorString(e));
    }
    *out = h;
    return VF_OK;
}

void vf_dyn_destroy(vf_dyn* h)
{
    if (!h) return;
    if (h->ctl) (void)hipFree(h->ctl);
    delete h;
}

int vf_dyn_bind(vf_dyn* h, float* slab, float* queue, const float* klin, const float* kquad)
{
    if (!h || !slab) return vf::fail(VF_EINVAL, "vf_dyn_bind: null handle or slab");
    if ((h->cfg.delay_steps > 0) != (queue != nullptr))
        return vf::fail(VF_EINVAL, "vf_dyn_bind: queue must be given iff delay_steps > 0 (delay_steps=%d)", h->cfg.delay_steps);
    if ((klin == nullptr) != (kquad == nullptr))
        return vf::fail(VF_EINVAL, "vf_dyn_bind: klin and kquad must be given together");
    h->S = slab;
    h->Q = queue;
    h->klin = klin;
    h->kquad = kquad;
    return VF_OK;
}

int vf_dyn_step(vf_dyn* h, const float* action, float* state_out, vf_stream_t stream)
{
    if (!h || !action) return vf::fail(VF_EINVAL, "vf_dyn_step: null handle or action");
    if (!h->S) return vf::fail(VF_ESTATE, "vf_dyn_step: vf_dyn_bind has not been called");
    return launch_step(h, action, state_out, vf::as_stream(stream));
}

int vf_dyn_reset(vf_dyn* h, const int32_t* idx, int32_t k, const float* pos, const float* quat, const float* vel,
                 const float* omg, const float* mot, const float* thr, const float* t, const float* t_rand,
                 vf_stream_t stream)
{
    if (!h) return vf::fail(VF_EINVAL, "vf_dyn_reset: null handle");
    if (!h->S) return vf::fail(VF_ESTATE, "vf_dyn_reset: vf_dyn_bind has not been called");
    const int n = idx ? k : h->N;
    if (n < 0) return vf::fail(VF_EINVAL, "vf_dyn_reset: k < 0");
    hipStream_t st = vf::as_stream(stream);
    if (!idx) VF_HIP(hipMemsetAsync(h->ctl, 0, 2 * sizeof(int), st));
    if (n == 0) return VF_OK;
    vf::ResetArgs r{h->N, n, h->S, h->Q, idx, pos, quat, vel, omg, mot, thr, t, t_rand};
    hipLaunchKernelGGL(vf::k_dyn_reset, dim3(vf::blocks_for(n)), dim3(vf::kBlock), 0, st, h->cfg, r);
    VF_HIP(hipGetLastError());
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
