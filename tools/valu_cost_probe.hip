// r02 probe: issue cost (shader cycles per wave64 instruction) of the VALU ops the fused control interval is made of,
// for ONE wave per SIMD: 8-way independent streams (throughput) and a dependent chain (latency).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_cost_probe.hip -o tools/valu_cost_probe && tools/valu_cost_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

// 8 independent instructions, repeated 64 times per loop trip = 512 instructions per trip
#define INDEP8(OP, A)                                                                                       \
    OP " %0, %0, " A "\n" OP " %1, %1, " A "\n" OP " %2, %2, " A "\n" OP " %3, %3, " A "\n" OP " %4, %4, " A "\n" \
    OP " %5, %5, " A "\n" OP " %6, %6, " A "\n" OP " %7, %7, " A "\n"

template <int KIND>
__global__ void probe(float* out, unsigned long long* cyc, int trips, float seed)
{
    float r[8];
    float2 p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        r[k] = seed + 0.001f * (threadIdx.x + k);
        p[k] = make_float2(r[k], r[k] + 0.5f);
    }
    const float m = 1.0000001f;
    const float2 m2 = make_float2(m, m);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < trips; ++t) {
        if constexpr (KIND == 0) {  // v_mul_f32 independent
            asm volatile(REP64(INDEP8("v_mul_f32", "%8"))
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                         : "v"(m));
        } else if constexpr (KIND == 1) {  // v_mul_f32 dependent chain
            asm volatile(REP64(REP4("v_mul_f32 %0, %0, %1\n") REP4("v_mul_f32 %0, %0, %1\n")) : "+v"(r[0]) : "v"(m));
        } else if constexpr (KIND == 2) {  // v_pk_mul_f32 independent
            asm volatile(REP64(INDEP8("v_pk_mul_f32", "%8"))
                         : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
                         : "v"(m2));
        } else if constexpr (KIND == 3) {  // v_pk_mul_f32 dependent
            asm volatile(REP64(REP4("v_pk_mul_f32 %0, %0, %1\n") REP4("v_pk_mul_f32 %0, %0, %1\n")) : "+v"(p[0]) : "v"(m2));
        } else if constexpr (KIND == 4) {  // v_fma_f32 independent
            asm volatile(REP64("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                               "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n")
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                         : "v"(m));
        } else if constexpr (KIND == 5) {  // v_pk_fma_f32 independent
            asm volatile(REP64("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                               "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n")
                         : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
                         : "v"(m2));
        } else if constexpr (KIND == 6) {  // v_rcp_f32 independent
            asm volatile(REP64("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                               "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
        } else if constexpr (KIND == 7) {  // v_sqrt_f32 independent
            asm volatile(REP64("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                               "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n")
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
        } else if constexpr (KIND == 8) {  // v_div_scale_f32 independent (writes vcc)
            asm volatile(REP64("v_div_scale_f32 %0, vcc, %0, %8, %0\n v_div_scale_f32 %1, vcc, %1, %8, %1\n"
                               "v_div_scale_f32 %2, vcc, %2, %8, %2\n v_div_scale_f32 %3, vcc, %3, %8, %3\n"
                               "v_div_scale_f32 %4, vcc, %4, %8, %4\n v_div_scale_f32 %5, vcc, %5, %8, %5\n"
                               "v_div_scale_f32 %6, vcc, %6, %8, %6\n v_div_scale_f32 %7, vcc, %7, %8, %7\n")
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                         : "v"(m)
                         : "vcc");
        } else if constexpr (KIND == 9) {  // v_div_fmas_f32 independent (reads vcc)
            asm volatile(REP64("v_div_fmas_f32 %0, %0, %8, %8\n v_div_fmas_f32 %1, %1, %8, %8\n v_div_fmas_f32 %2, %2, %8, %8\n"
                               "v_div_fmas_f32 %3, %3, %8, %8\n v_div_fmas_f32 %4, %4, %8, %8\n v_div_fmas_f32 %5, %5, %8, %8\n"
                               "v_div_fmas_f32 %6, %6, %8, %8\n v_div_fmas_f32 %7, %7, %8, %8\n")
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                         : "v"(m)
                         : "vcc");
        } else if constexpr (KIND == 10) {  // v_div_fixup_f32 independent
            asm volatile(REP64("v_div_fixup_f32 %0, %0, %8, %8\n v_div_fixup_f32 %1, %1, %8, %8\n v_div_fixup_f32 %2, %2, %8, %8\n"
                               "v_div_fixup_f32 %3, %3, %8, %8\n v_div_fixup_f32 %4, %4, %8, %8\n v_div_fixup_f32 %5, %5, %8, %8\n"
                               "v_div_fixup_f32 %6, %6, %8, %8\n v_div_fixup_f32 %7, %7, %8, %8\n")
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                         : "v"(m));
        } else if constexpr (KIND == 11) {  // v_mov_b32 independent
            asm volatile(REP64("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                               "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                         : "v"(m));
        } else if constexpr (KIND == 12) {  // a full IEEE division per iteration (what `a / b` compiles to), 8 independent
#pragma unroll
            for (int u = 0; u < 64; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) r[k] = r[k] / m;
        } else if constexpr (KIND == 13) {  // the FMA-corrected reciprocal product (5 ops), 8 independent
            const float y = 1.0f / m;
#pragma unroll
            for (int u = 0; u < 64; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float q0 = r[k] * y;
                    float e0 = __builtin_fmaf(-m, q0, r[k]);
                    float q1 = __builtin_fmaf(e0, y, q0);
                    float e1 = __builtin_fmaf(-m, q1, r[k]);
                    r[k] = __builtin_fmaf(e1, y, q1);
                }
        } else if constexpr (KIND == 14) {  // IEEE sqrt, 8 independent
#pragma unroll
            for (int u = 0; u < 64; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) r[k] = __builtin_sqrtf(r[k]) + 1.0f;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += r[k] + p[k].x + p[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_trip, int blocks)
{
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, blocks * 64 * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    const int trips = 20;
    probe<KIND><<<blocks, 64>>>(out, cyc, trips, 1.0f);
    hipDeviceSynchronize();
    probe<KIND><<<blocks, 64>>>(out, cyc, trips, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-44s waves=%5d  %.2f cycles per %s\n", name, blocks, s / blocks / trips / per_trip, KIND >= 12 ? "operation" : "instruction");
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    for (int blocks : {1, 1024, 2048, 4096}) {
        run<0>("v_mul_f32 x8 independent", 512, blocks);
        run<1>("v_mul_f32 dependent chain", 512, blocks);
        run<2>("v_pk_mul_f32 x8 independent", 512, blocks);
        run<3>("v_pk_mul_f32 dependent chain", 512, blocks);
        run<4>("v_fma_f32 x8 independent", 512, blocks);
        run<5>("v_pk_fma_f32 x8 independent", 512, blocks);
        run<6>("v_rcp_f32 x8 independent", 512, blocks);
        run<7>("v_sqrt_f32 x8 independent", 512, blocks);
        run<8>("v_div_scale_f32 x8 independent", 512, blocks);
        run<9>("v_div_fmas_f32 x8 independent", 512, blocks);
        run<10>("v_div_fixup_f32 x8 independent", 512, blocks);
        run<11>("v_mov_b32 x8 independent", 512, blocks);
        run<12>("IEEE a / b (compiler expansion) x8 indep.", 512, blocks);
        run<13>("twice-corrected reciprocal product x8", 512, blocks);
        run<14>("IEEE sqrt (+1 add) x8 independent", 512, blocks);
    }
    return 0;
}
