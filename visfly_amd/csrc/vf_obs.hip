// vf_obs.hip -- observation rows that are not a per-agent function of the step alone.
//
// RacingEnv2 (envs/RacingEnv.py:218-267): "state" = [(next `n_next` gates - p) / max_sense_radius, q, v / 10, w / 10], "gate" = the index
// used.  WHICH gate index the rows returned by step() use depends on the whole batch (RacingEnv notes in visfly_amd/envs/tasks.py: the
// reference refreshes the observation before it advances the gate of an agent that just passed one -- unless SOME agent ended its
// episode in the step, which rebuilds every agent's observation with the advanced gates), so the rows cannot come out of the step kernel's
// epilogue, which sees one agent.  They are one launch behind it: the step kernel's compacted done list says whether any episode ended
// (vf_env_out.done_count), this kernel picks the index and forms the rows -- instead of the dozen torch launches of the host version.
#include "vf_common.hpp"

namespace vf {

constexpr int kRaceMaxNext = 4;
struct RaceGates {
    float g[VF_MAX_GATES][3];
};

__global__ __launch_bounds__(kBlock) void k_race_obs(const float* __restrict__ raw, const int* __restrict__ gate, const int* __restrict__ gate_prev,
                                                     const int* __restrict__ done_count, int mode, const RaceGates gt, int n_gates, int n_next,
                                                     float radius, float* __restrict__ state, int* __restrict__ gate_out, int N)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    // mode 0: the current index; 1: the index at the start of the step; 2: the current one iff some episode ended in the step
    const bool cur = mode == 0 || (mode == 2 && *done_count > 0);
    const int g0 = cur ? gate[i] : gate_prev[i];
    const float* r = raw + 13 * (size_t)i;
    const float p[3] = {r[0], r[1], r[2]};
    const int W = 3 * n_next + 10;
    float* o = state + (size_t)W * i;
    for (int k = 0; k < n_next; ++k) {
        const int gi = (g0 + k) % n_gates;                                       // RacingEnv.py:254
#pragma unroll
        for (int c = 0; c < 3; ++c) o[3 * k + c] = (gt.g[gi][c] - p[c]) / radius;     // :255-257 (IEEE division, as torch's CPU kernels)
    }
    float* t = o + 3 * n_next;
#pragma unroll
    for (int c = 0; c < 4; ++c) t[c] = r[3 + c];                                 // :258
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        t[4 + c] = r[7 + c] / 10.0f;                                             // :259-260
        t[7 + c] = r[10 + c] / 10.0f;
    }
    if (gate_out) gate_out[i] = g0;
}

}  // namespace vf

extern "C" int vf_race_obs(const float* raw, const int32_t* gate, const int32_t* gate_prev, const int32_t* done_count, int32_t mode,
                           const float* gates_host, int32_t n_gates, int32_t n_next, float radius, float* state, int32_t* gate_out, int32_t N,
                           vf_stream_t stream)
{
    using namespace vf;
    if (!raw || !gate || !gates_host || !state || N < 1) return fail(VF_EINVAL, "vf_race_obs: null argument");
    if (n_gates < 1 || n_gates > VF_MAX_GATES || n_next < 1 || n_next > kRaceMaxNext || !(radius > 0.0f))
        return fail(VF_EINVAL, "vf_race_obs: n_gates in [1, %d], n_next in [1, %d], radius > 0", VF_MAX_GATES, kRaceMaxNext);
    if (mode < 0 || mode > 2 || (mode != 0 && !gate_prev) || (mode == 2 && !done_count))
        return fail(VF_EINVAL, "vf_race_obs: mode 1 / 2 need gate_prev, mode 2 needs done_count");
    RaceGates gt{};
    for (int k = 0; k < n_gates; ++k)
        for (int c = 0; c < 3; ++c) gt.g[k][c] = gates_host[3 * k + c];
    hipLaunchKernelGGL(k_race_obs, dim3(blocks_for(N)), dim3(kBlock), 0, as_stream(stream), raw, gate, gate_prev, done_count, mode, gt, n_gates,
                       n_next, radius, state, gate_out, N);
    VF_HIP(hipGetLastError());
    return VF_OK;
}
