// What would forming the weight gradients INSIDE the PPO chain kernel cost?  (DESIGN "PPO optimiser step".)
//
// Skeleton of k_ppo_update_chain as tools/mfma_chain_probe.hip has it (1408 dependent-quartet MFMAs per 32-row wave, one float4
// weight load per 4 MFMAs through a ring of 8, bias + ReLU between layers) in three forms:
//   k_chain_stores   the shipped shape: activations / masked gradients trickled out to HBM for k_mlp_wgrad (one float4 per 2 items)
//   k_chain_nostore  the same without those stores (bound: nothing leaves the wave)
//   k_chain_fused    no stores; 4 waves = one 128-row workgroup; at seven stage points of the reverse half every wave parks the
//                    stage's X / dZ tiles TRANSPOSED in LDS ([feature][row], ds_write_b32), barrier, then forms its share of the
//                    stage's dW tiles over all 128 rows (v_mfma_f32_32x32x2_f32, both operands ds_read_b128 = four steps per
//                    read), stores one partial per workgroup, barrier.  Stage table = the StateTarget network's layers:
//                    13 tile pairs x 64 MFMAs = 832 MFMAs per wave, 44 tile writes per wave, 208 partial stores per wave.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/wgrad_fused_probe tools/wgrad_fused_probe.hip && tools/wgrad_fused_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kItems = 352, kDepth = 8, kLayerItems = 64, kImageFloat4 = 352 * 64;
constexpr int kLd = 132;                       // floats per feature row of a parked tile: 128 rows + 4 (bank spread of the b128 reads)
constexpr int kTileFloats = 32 * kLd;          // one 32-feature tile of 128 rows

struct Stage { int item, n_write, n_pairs, share; };   // after chain item `item`: tiles parked per wave, dW tile pairs per wave
// heads | pi2+vf2 | pi1+vf1 | feat_a | feat_b | e1_a | e1_b
__host__ __device__ constexpr Stage stage(int s)
{
    switch (s) {
    case 0: return Stage{196, 6, 1, 0};
    case 1: return Stage{216, 8, 2, 1};
    case 2: return Stage{244, 8, 4, 1};
    case 3: return Stage{276, 6, 2, 1};
    case 4: return Stage{304, 6, 2, 1};
    case 5: return Stage{328, 5, 1, 0};
    default: return Stage{351, 5, 1, 0};
    }
}
constexpr int kStages = 7;
constexpr int kPairsPerWave = 13;

template <int I>
__device__ __forceinline__ float4 wload(const float4* w, int lane) { return w[(I % kItems) * 64 + lane]; }

template <int S>
__device__ __forceinline__ void wgrad_stage(float* lds, f32x16 (&t)[8], float* part, int lane, int wave, int& pair0)
{
    constexpr Stage st = stage(S);
    const int m = lane & 31, h = lane >> 5, c = lane & 31, kk = lane >> 5;
    // park: feature n = 8 (r >> 2) + 4 h + (r & 3) of row 32 wave + m
#pragma unroll
    for (int tw = 0; tw < st.n_write; ++tw) {
        float* tile = lds + tw * kTileFloats + 32 * wave + m;
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[(8 * (r >> 2) + 4 * h + (r & 3)) * kLd] = t[tw & 7][r];
    }
    __syncthreads();
    // this wave's tile pairs: A tile ta, B tiles tb .. (share: two pairs on one A tile)
#pragma unroll
    for (int p = 0; p < st.n_pairs; p += (st.share ? 2 : 1)) {
        const int ta = (wave + p) % st.n_write, tb0 = (wave + p + 1) % st.n_write, tb1 = (wave + p + 2) % st.n_write;
        const float* pa = lds + ta * kTileFloats + c * kLd + 4 * kk;
        const float* pb0 = lds + tb0 * kTileFloats + c * kLd + 4 * kk;
        const float* pb1 = lds + tb1 * kTileFloats + c * kLd + 4 * kk;
        f32x16 acc0{0}, acc1{0};
        float4 a = *reinterpret_cast<const float4*>(pa), b0 = *reinterpret_cast<const float4*>(pb0), b1 = b0;
        if constexpr (st.share) b1 = *reinterpret_cast<const float4*>(pb1);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float4 an = a, b0n = b0, b1n = b1;
            if (j + 1 < 16) {
                an = *reinterpret_cast<const float4*>(pa + 8 * (j + 1));
                b0n = *reinterpret_cast<const float4*>(pb0 + 8 * (j + 1));
                if constexpr (st.share) b1n = *reinterpret_cast<const float4*>(pb1 + 8 * (j + 1));
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
            if constexpr (st.share) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
            if constexpr (st.share) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
            if constexpr (st.share) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
            if constexpr (st.share) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
            a = an; b0 = b0n; b1 = b1n;
            __builtin_amdgcn_sched_barrier(0);
        }
        float* o = part + (size_t)(pair0 + p) * 1024 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r * 64] = acc0[r];
        if constexpr (st.share) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[1024 + r * 64] = acc1[r];
        }
    }
    pair0 += st.n_pairs;
    __syncthreads();
}

template <int I, int MODE>     // MODE 0: trickled stores, 1: no stores, 2: fused weight gradients
__device__ __forceinline__ void items32(const float4* w, float4* out, f32x16 (&t)[8], float4 (&ring)[kDepth], int lane, int row,
                                        float* lds, float* part, int wave, int& pair0)
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
        if constexpr (MODE == 0 && (I & 1) == 0 && layer >= 1) {
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
        if constexpr (MODE == 2) {
            if constexpr (I == stage(0).item) wgrad_stage<0>(lds, t, part, lane, wave, pair0);
            if constexpr (I == stage(1).item) wgrad_stage<1>(lds, t, part, lane, wave, pair0);
            if constexpr (I == stage(2).item) wgrad_stage<2>(lds, t, part, lane, wave, pair0);
            if constexpr (I == stage(3).item) wgrad_stage<3>(lds, t, part, lane, wave, pair0);
            if constexpr (I == stage(4).item) wgrad_stage<4>(lds, t, part, lane, wave, pair0);
            if constexpr (I == stage(5).item) wgrad_stage<5>(lds, t, part, lane, wave, pair0);
            if constexpr (I == stage(6).item) wgrad_stage<6>(lds, t, part, lane, wave, pair0);
        }
        items32<I + 1, MODE>(w, out, t, ring, lane, row, lds, part, wave, pair0);
    }
}

template <int MODE>
__global__ __launch_bounds__(MODE == 2 ? 256 : 64) void k_chain(const float4* w, const float* x, float4* out, float* partials, int M)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = (blockIdx.x * (MODE == 2 ? 4 : 1) + wave) * 32 + (lane & 31);
    f32x16 t[8];
    float4 ring[kDepth];
#pragma unroll
    for (int i = 0; i < kDepth; ++i) ring[i] = w[i * 64 + lane];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) t[a][r] = x[(size_t)(row % M) * 64 + 16 * a + r];
    int pair0 = 0;
    float* part = partials + ((size_t)blockIdx.x * 4 + wave) * kPairsPerWave * 1024;
    items32<0, MODE>(w, out, t, ring, lane, row, lds, part, wave, pair0);
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < 8; ++a) s += t[a][0];
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
    float *x, *part;
    CK(hipMalloc(&w, kImageFloat4 * sizeof(float4)));
    CK(hipMalloc(&out, (size_t)(M + 64) * 352 * sizeof(float4)));
    CK(hipMalloc(&x, (size_t)M * 64 * sizeof(float)));
    CK(hipMalloc(&part, (size_t)1024 * kPairsPerWave * 1024 * sizeof(float)));
    std::vector<float> hw(kImageFloat4 * 4, 0.01f), hx((size_t)M * 64, 0.5f);
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    const int lds_bytes = 8 * kTileFloats * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_chain<0>)));
    printf("k_chain<stores>: %d regs\n", fa.numRegs);
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_chain<2>)));
    printf("k_chain<fused>: %d regs, %d B LDS (dynamic %d)\n", fa.numRegs, (int)fa.sharedSizeBytes, lds_bytes);
    for (int rep = 0; rep < 3; ++rep) {
        const float a = time_us([&] { hipLaunchKernelGGL(k_chain<0>, dim3(M / 32), dim3(64), 0, 0, w, x, out, part, M); }, 50);
        const float b = time_us([&] { hipLaunchKernelGGL(k_chain<1>, dim3(M / 32), dim3(64), 0, 0, w, x, out, part, M); }, 50);
        const float c = time_us([&] { hipLaunchKernelGGL(k_chain<2>, dim3(M / 128), dim3(256), lds_bytes, 0, w, x, out, part, M); }, 50);
        printf("25600 rows: stores to HBM %.1f us | no stores %.1f us | fused weight gradients (200 workgroups of 4 waves) %.1f us\n", a, b, c);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
