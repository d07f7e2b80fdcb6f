// How does the cost of a DEPENDENT fp32 MFMA chain depend on how many waves share the CU / the SIMD?  (r05: the split chain kernels --
// two co-resident half-chain waves per SIMD -- gained nothing over one full-chain wave per SIMD; profiles/r05_chain_split.txt.)
// Every wave runs n items of 4 x v_mfma_f32_32x32x2_f32 on one accumulator (the chain kernels' item), B operand = a register of another
// tile, A operand = a float4 from a ring of 8 loads out of a 352 KB image (LOADS) or a register (no memory at all), and reports its
// shader cycles + where it ran.  Grid sizes put 1, 2, 4 (one per SIMD), 8 (two per SIMD) waves on every CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_occupancy_probe tools/mfma_occupancy_probe.hip && tools/mfma_occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kDepth = 8, kImageItems = 352;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__shared__ v4f lds_w[8][64];

template <int LOADS, int ACCS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_probe(const float4* __restrict__ w, const float* __restrict__ x, float* out,
                                                                                         unsigned long long* rec, int iters)
{
    const int lane = threadIdx.x;
    f32x16 t[8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) t[a][r] = x[(blockIdx.x * 64 + lane) % 4096 + 7 * (16 * a + r)];
    float4 ring[kDepth];
#pragma unroll
    for (int i = 0; i < kDepth; ++i) ring[i] = w[i * 64 + lane];
    // buffer descriptor over the image (raw buffer, 32-bit offsets)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(w), 0, kImageItems * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t osrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, 8u << 20, 0x00020000);
    if (LOADS == 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) lds_w[i][lane] = v4f{ring[i].x, ring[i].y, ring[i].z, ring[i].w};
        __syncthreads();
    }
    const unsigned voff = lane * 16u;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    int item = kDepth;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < kDepth; ++j) {
            const float4 wv = ring[j];
            if (LOADS == 1) {                     // what the chain kernels do today: pointer arithmetic -> 64-bit VGPR address
                ring[j] = w[(item % kImageItems) * 64 + lane];
                ++item;
            } else if (LOADS == 2) {              // scalar base + 32-bit lane offset
                const char* base = reinterpret_cast<const char*>(w) + (size_t)(item % kImageItems) * 1024;
                unsigned long long bb = reinterpret_cast<unsigned long long>(base);
                bb = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)bb) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bb >> 32)) << 32);
                ring[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(bb) + voff);
                ++item;
            } else if (LOADS == 3) {              // buffer load: descriptor + lane offset + scalar offset
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (item % kImageItems) * 1024, 0);
                ring[j] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
                ++item;
            } else if (LOADS == 4) {              // LDS read (someone else filled it)
                const v4f v = lds_w[j][lane];
                ring[j] = make_float4(v[0], v[1], v[2], v[3]);
            } else if (LOADS == 6) {              // a STORE per item instead (the chain kernels' trickled copies): 64-bit vaddr
                float4* o = reinterpret_cast<float4*>(out) + ((size_t)(item & 63) * 4096 + blockIdx.x * 64 + lane) % (4096 * 16);
                *o = wv;
                ++item;
            } else if (LOADS == 7) {              // ... scalar base + 32-bit lane offset
                char* base = reinterpret_cast<char*>(out) + (size_t)(item & 63) * 65536;
                unsigned long long bb = reinterpret_cast<unsigned long long>(base);
                bb = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)bb) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bb >> 32)) << 32);
                *reinterpret_cast<float4*>(reinterpret_cast<char*>(bb) + (voff + (blockIdx.x & 63) * 1024u)) = wv;
                ++item;
            } else if (LOADS == 8) {              // ... buffer store
                __builtin_amdgcn_raw_buffer_store_b128(v4u{__float_as_uint(wv.x), __float_as_uint(wv.y), __float_as_uint(wv.z), __float_as_uint(wv.w)}, osrc,
                                                       (int)(voff + (blockIdx.x & 63) * 1024u), (item & 63) * 65536, 0);
                ++item;
            } else if (LOADS == 5) {              // one dword per lane instead of four (is the cost the address or the data?)
                ring[j].x = reinterpret_cast<const float*>(w)[(item % kImageItems) * 256 + lane];
                ++item;
            }
            f32x16& acc = t[4 + j % ACCS];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float b = t[j / 2][(4 * j) % 16 + k];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k == 0 ? wv.x : k == 1 ? wv.y : k == 2 ? wv.z : wv.w, b, acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
#pragma unroll
    for (int a = 4; a < 8; ++a) s += t[a][lane & 15];
    out[blockIdx.x * 64 + lane] = s;
    if (lane == 0) {
        rec[4 * blockIdx.x + 0] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF) << 32);
        rec[4 * blockIdx.x + 1] = c1 - c0;
        rec[4 * blockIdx.x + 2] = r0;
        rec[4 * blockIdx.x + 3] = r1;
    }
}

template <int LOADS, int ACCS>
void run(const char* tag, int waves, const float4* w, const float* x, float* out, unsigned long long* rec, int iters)
{
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_probe<LOADS, ACCS>), dim3(waves), dim3(64), 0, 0, w, x, out, rec, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(4 * (size_t)waves);
    hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost);
    std::map<unsigned long long, int> per_simd, per_cu;
    double cyc = 0, cmax = 0;
    unsigned long long r0 = ~0ull, r1 = 0;
    for (int i = 0; i < waves; ++i) {
        const unsigned long long hw = h[4 * i] & 0xFFFFFFFFull, xcc = h[4 * i] >> 32;
        const unsigned long long cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
        per_cu[cu]++;
        per_simd[(cu << 2) | ((hw >> 4) & 3)]++;
        cyc += (double)h[4 * i + 1];
        cmax = std::max(cmax, (double)h[4 * i + 1]);
        r0 = std::min(r0, h[4 * i + 2]);
        r1 = std::max(r1, h[4 * i + 3]);
    }
    int mx_s = 0, mx_c = 0;
    for (auto& p : per_simd) mx_s = std::max(mx_s, p.second);
    for (auto& p : per_cu) mx_c = std::max(mx_c, p.second);
    const double n_mfma = 32.0 * iters;
    printf("%-34s waves %5d on %4zu CUs / %4zu SIMDs (max %d per CU, %d per SIMD): %.1f cycles per MFMA per wave (max %.1f), span %.1f us, %.2f GHz, %.1f TF/s\n",
           tag, waves, per_cu.size(), per_simd.size(), mx_c, mx_s, cyc / waves / n_mfma, cmax / n_mfma, (r1 - r0) / 100.0,
           cyc / waves / ((r1 - r0) * 10.0), waves * n_mfma * 4096.0 / ((r1 - r0) * 1e-8) / 1e12);
}

int main()
{
    const int iters = 88;       // 88 x 8 items x 4 = 2 816 MFMAs per wave (two PPO update chains)
    float4* w;
    float *x, *out;
    unsigned long long* rec;
    hipMalloc(&w, kImageItems * 64 * sizeof(float4));
    hipMalloc(&x, 8192 * sizeof(float));
    hipMalloc(&out, 8u << 20);      // 8 MiB: the store variants spread 64 items x 64 KiB
    hipMalloc(&rec, 4096 * 4 * 8);
    std::vector<float> hw(kImageItems * 64 * 4), hx(8192);
    srand(1);
    for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.05f;
    for (auto& v : hx) v = (rand() / (float)RAND_MAX - 0.5f);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    for (int waves : {256, 1024}) {
        run<0, 1>("no loads, 1 accumulator", waves, w, x, out, rec, iters);
        run<0, 4>("no loads, 4 accumulators in turn", waves, w, x, out, rec, iters);
        run<1, 4>("ring: global, 64-bit vaddr", waves, w, x, out, rec, iters);
        run<2, 4>("ring: global, saddr + voffset", waves, w, x, out, rec, iters);
        run<3, 4>("ring: buffer_load offen + soffset", waves, w, x, out, rec, iters);
        run<4, 4>("ring: ds_read_b128", waves, w, x, out, rec, iters);
        run<5, 4>("ring: global dword", waves, w, x, out, rec, iters);
        run<6, 4>("store per item: global 64-bit vaddr", waves, w, x, out, rec, iters);
        run<7, 4>("store per item: global saddr + voff", waves, w, x, out, rec, iters);
        run<8, 4>("store per item: buffer_store offen", waves, w, x, out, rec, iters);
    }
    return 0;
}
