// probe: shader cycles (s_memtime) and wall time of ONE wave running the fused control interval
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "vf_dyn_device.hpp"
using namespace vf;

__global__ void probe(const vf_dyn_cfg c, float* S, int G, const float4* act, unsigned long long* out, int reps)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Agent s; Spares sp;
    load_agent(S, G, i, s, sp);
    float4 a4 = act[i];
    float a[4] = {a4.x, a4.y, a4.z, a4.w};
    float kl[3] = {c.k_lin[0], c.k_lin[1], c.k_lin[2]}, kq[3] = {c.k_quad[0], c.k_quad[1], c.k_quad[2]};
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long w0 = wall_clock64();
    for (int r = 0; r < reps; ++r) control_interval<VF_ACT_BODYRATE, VF_INT_EULER, true>(c, s, a, kl, kq);
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long w1 = wall_clock64();
    store_agent(S, G, i, s, sp);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = w1 - w0; }
}

int main(int argc, char** argv)
{
    vf_dyn_cfg c; memset(&c, 0, sizeof(c));
    FILE* f = fopen(argv[1], "rb"); fread(&c, sizeof(c), 1, f); fclose(f);
    const int G = 11;
    for (int blocks : {1, 256, 1024, 4096}) {
        const int N = blocks * 64;
        float* S; float4* act; unsigned long long* out;
        hipMalloc(&S, (size_t)N * G * 16); hipMalloc(&act, (size_t)N * 16); hipMalloc(&out, blocks * 16);
        std::vector<float> h((size_t)N * G * 4, 0.f);
        for (int i = 0; i < N; ++i) {  // hover state
            auto at = [&](int g, int k) -> float& { return h[(((size_t)(i >> 6) * G + g) * 64 + (i & 63)) * 4 + k]; };
            at(0, 1) = 1.f; at(0, 3) = 1.5f; at(1, 0) = 1.f;
            for (int k = 0; k < 4; ++k) { at(4, k) = c.w_init; at(5, k) = c.T_init; }
        }
        hipMemcpy(S, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> ha((size_t)N * 4);
        for (int i = 0; i < N; ++i) { ha[4 * i] = -0.3333f + 0.001f * (i % 7); ha[4 * i + 1] = 0.01f; ha[4 * i + 2] = -0.01f; ha[4 * i + 3] = 0.f; }
        hipMemcpy(act, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
        for (int reps : {1, 4}) {
            probe<<<blocks, 64>>>(c, S, G, act, out, reps);
            hipDeviceSynchronize();
            probe<<<blocks, 64>>>(c, S, G, act, out, reps);
            hipDeviceSynchronize();
            std::vector<unsigned long long> o(blocks * 2);
            hipMemcpy(o.data(), out, blocks * 16, hipMemcpyDeviceToHost);
            double cy = 0, wl = 0;
            for (int b = 0; b < blocks; ++b) { cy += o[2 * b]; wl += o[2 * b + 1]; }
            cy /= blocks; wl /= blocks;
            printf("waves=%5d (64-thread blocks) reps=%d: %.0f shader cycles, %.2f us wall (100MHz ctr) per wave -> %.0f cyc/interval, %.2f GHz\n",
                   blocks, reps, cy, wl / 100.0, cy / reps, cy / (wl * 10.0));
        }
        hipFree(S); hipFree(act); hipFree(out);
    }
    return 0;
}
