// shader clock under MFMA load: every wave issues N dependent v_mfma_f32_32x32x2_f32 (64 pipe cycles each) and reports
// shader-clock ticks (s_memtime) and wall-clock ticks (100 MHz) -> effective MHz and cycles per MFMA
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_clock_probe.hip -o tools/mfma_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ void k(int n, float* out, long long* t)
{
    f32x16 a0 = {0}, a1 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    const long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < n; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 64 + threadIdx.x] = a0[0] + a1[3];
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = w1 - w0; }
}
int main()
{
    const int n = 20000;
    for (int blocks : {1, 256, 1024, 2048}) {
        float* out; long long* t;
        hipMalloc(&out, blocks * 64 * 4); hipMalloc(&t, blocks * 16);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, n, out, t);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, n, out, t);
        hipDeviceSynchronize();
        long long h[2];
        hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("waves %5d: shader ticks %lld wall ticks(100MHz) %lld -> %.0f MHz, %.1f shader cycles per MFMA\n", blocks, h[0], h[1],
               h[0] / (h[1] / 100.0), (double)h[0] / (2.0 * n));
    }
    return 0;
}
