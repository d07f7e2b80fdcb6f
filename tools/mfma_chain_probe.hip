// Would the register-chained MLP kernels (vf_mlp_chain.hip) run faster with 16 rows per wave (v_mfma_f32_16x16x4_f32, two
// waves per SIMD) than with 32 rows per wave (v_mfma_f32_32x32x2_f32, one wave per SIMD)?  Same structure as the real kernel:
// an "item" = one float4 weight load (ring of 8 in flight, streamed from a 352 KB image in L2) feeding 4 MFMAs whose B operand
// is an accumulator register of the previous layer; bias + ReLU between layers; a float4 store per two items.
// 1408 MFMAs per wave either way (= the PPO update chain of the StateTarget network); 25 600 rows.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_chain_probe tools/mfma_chain_probe.hip && tools/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kItems = 352, kDepth = 8, kLayerItems = 64, kImageFloat4 = 352 * 64;   // 352 KB of packed weights

template <int I>
__device__ __forceinline__ float4 wload(const float4* w, int lane) { return w[(I % kItems) * 64 + lane]; }

// ---- 32 rows per wave ------------------------------------------------------------------------------------------------------
template <int I>
__device__ __forceinline__ void items32(const float4* w, float4* out, f32x16 (&t)[8], float4 (&ring)[kDepth], int lane, int row)
{
    if constexpr (I < kItems) {
        constexpr int layer = I / kLayerItems, local = I % kLayerItems, gq = local / 4, a = local % 4;
        constexpr int in0 = (layer & 1) * 4, out0 = ((layer + 1) & 1) * 4;
        const float4 wv = ring[I % kDepth];
        if constexpr (I + kDepth < kItems) ring[I % kDepth] = wload<I + kDepth>(w, lane);
        f32x16& acc = t[out0 + a];
        if constexpr (gq == 0) acc = f32x16{0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float b = t[in0 + gq / 4][4 * (gq % 4) + j];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(j == 0 ? wv.x : j == 1 ? wv.y : j == 2 ? wv.z : wv.w, b, acc, 0, 0, 0);
        }
        if constexpr ((I & 1) == 0 && layer >= 1) {   // deferred store of the previous layer's output
            constexpr int i = (local / 2) % 16, ta = i / 4, q = i % 4;
            const f32x16& y = t[in0 + ta];
            out[(size_t)(row % 25600) * 352 + (I / 2)] = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (local == kLayerItems - 1) {
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int r = 0; r < 16; ++r) t[out0 + x][r] = fmaxf(t[out0 + x][r] + wv.x, 0.0f);
        }
        items32<I + 1>(w, out, t, ring, lane, row);
    }
}

__global__ __launch_bounds__(64) void k_chain32(const float4* w, const float* x, float4* out, int M)
{
    const int lane = threadIdx.x, row = blockIdx.x * 32 + (lane & 31);
    f32x16 t[8];
    float4 ring[kDepth];
#pragma unroll
    for (int i = 0; i < kDepth; ++i) ring[i] = w[i * 64 + lane];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) t[a][r] = x[(size_t)(row % M) * 64 + 16 * a + r];
    items32<0>(w, out, t, ring, lane, row);
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < 8; ++a) s += t[a][0];
    if (s == 12345.0f) out[0] = make_float4(s, s, s, s);
}

// ---- 16 rows per wave ------------------------------------------------------------------------------------------------------
// accumulator lane (row j = lane & 15, g = lane >> 4) holds features 4 g + r, r = 0..3, of a 16-feature tile; as the B operand of
// step r of the next layer it supplies k = g, i.e. feature 4 g + r -- the weights are packed in that order
template <int I>
__device__ __forceinline__ void items16(const float4* w, float4* out, f32x4 (&t)[16], float4 (&ring)[kDepth], int lane, int row)
{
    if constexpr (I < kItems) {
        constexpr int layer = I / kLayerItems, local = I % kLayerItems, gq = local / 8, a = local % 8;   // 8 out tiles x 8 float4 groups
        constexpr int in0 = (layer & 1) * 8, out0 = ((layer + 1) & 1) * 8;
        const float4 wv = ring[I % kDepth];
        if constexpr (I + kDepth < kItems) ring[I % kDepth] = wload<I + kDepth>(w, lane);
        f32x4& acc = t[out0 + a];
        if constexpr (gq == 0) acc = f32x4{0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float b = t[in0 + gq][j];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(j == 0 ? wv.x : j == 1 ? wv.y : j == 2 ? wv.z : wv.w, b, acc, 0, 0, 0);
        }
        if constexpr ((I & 3) == 0 && layer >= 1) {   // half as many stores per wave: 16 rows
            constexpr int ta = (local / 4) % 8;
            const f32x4& y = t[in0 + ta];
            out[(size_t)(row % 25600) * 352 + (I / 4)] = make_float4(y[0], y[1], y[2], y[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (local == kLayerItems - 1) {
#pragma unroll
            for (int x = 0; x < 8; ++x)
#pragma unroll
                for (int r = 0; r < 4; ++r) t[out0 + x][r] = fmaxf(t[out0 + x][r] + wv.x, 0.0f);
        }
        items16<I + 1>(w, out, t, ring, lane, row);
    }
}

template <int MINW>
__global__ __launch_bounds__(64, MINW) void k_chain16(const float4* w, const float* x, float4* out, int M)
{
    const int lane = threadIdx.x, row = blockIdx.x * 16 + (lane & 15);
    f32x4 t[16];
    float4 ring[kDepth];
#pragma unroll
    for (int i = 0; i < kDepth; ++i) ring[i] = w[i * 64 + lane];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[a][r] = x[(size_t)(row % M) * 64 + 4 * a + r];
    items16<0>(w, out, t, ring, lane, row);
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < 16; ++a) s += t[a][0];
    if (s == 12345.0f) out[0] = make_float4(s, s, s, s);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <class F>
float time_us(F launch, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / iters;
}

int main()
{
    const int M = 25600;
    float4 *w, *out;
    float* x;
    CK(hipMalloc(&w, kImageFloat4 * sizeof(float4)));
    CK(hipMalloc(&out, (size_t)(M + 64) * 352 * sizeof(float4)));
    CK(hipMalloc(&x, (size_t)M * 64 * sizeof(float)));
    std::vector<float> hw(kImageFloat4 * 4, 0.01f), hx((size_t)M * 64, 0.5f);
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_chain32)));
    printf("k_chain32: %d regs\n", fa.numRegs);
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_chain16<2>)));
    printf("k_chain16: %d regs\n", fa.numRegs);
    for (int rep = 0; rep < 2; ++rep) {
        const float a = time_us([&] { hipLaunchKernelGGL(k_chain32, dim3(M / 32), dim3(64), 0, 0, w, x, out, M); }, 50);
        const float b = time_us([&] { hipLaunchKernelGGL(k_chain16<2>, dim3(M / 16), dim3(64), 0, 0, w, x, out, M); }, 50);
        const float c = time_us([&] { hipLaunchKernelGGL(k_chain16<1>, dim3(M / 16), dim3(64), 0, 0, w, x, out, M); }, 50);
        const float d = time_us([&] { hipLaunchKernelGGL(k_chain32, dim3(1024), dim3(64), 0, 0, w, x, out, M); }, 50);
        const float e = time_us([&] { hipLaunchKernelGGL(k_chain16<2>, dim3(2048), dim3(64), 0, 0, w, x, out, M); }, 50);
        printf("25600 rows: 32 rows/wave (800 waves) %.1f us | 16 rows/wave (1600 waves, 2 per SIMD) %.1f us | 16 rows/wave launch_bounds(64,1) %.1f us\n", a, b, c);
        printf("full chip : 32 rows/wave x 1024 waves (32768 rows) %.1f us | 16 rows/wave x 2048 waves (32768 rows) %.1f us\n", d, e);
        const float f = time_us([&] { hipLaunchKernelGGL(k_chain32, dim3(512), dim3(64), 0, 0, w, x, out, M); }, 50);
        const float h = time_us([&] { hipLaunchKernelGGL(k_chain16<1>, dim3(1024), dim3(64), 0, 0, w, x, out, M); }, 50);
        printf("half chip : 32 rows/wave x 512 waves (16384 rows) %.1f us | 16 rows/wave x 1024 waves (16384 rows) %.1f us\n", f, h);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
