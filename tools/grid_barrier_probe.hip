// What does a device-wide meeting of ~1 024 lone waves cost on gfx950, and which part of it?  (round 6: the fused optimiser tail's first
// form -- every wave adds to ONE counter and polls ONE word -- took 110 us per meeting.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/grid_barrier_probe tools/grid_barrier_probe.hip && tools/grid_barrier_probe
// Variants (all: `waves` workgroups of 64 threads, launched back to back, mean launch time from HIP events):
//   0 empty kernel                         1 arrive only: atomicAdd on one word (agent scope)
//   2 arrive + all poll that word          3 arrive on one word, the last arrival writes one flag PER WAVE (64 B apart), each wave polls its own
//   4 arrive per XCD (XCC_ID), the XCD's last arrival adds to the global word and polls it, then raises the XCD's flag; the others poll that
//   5 no atomics: each wave raises its own arrival flag, wave 0 polls all of them and raises one release flag per wave
//   6 as 2 with s_sleep 8 between polls    7 as 4 with per-wave release flags written by the XCD's leader
//   8 payload + release fence only, no meeting      9 atomicAdd only, no fence, no payload
//   10 as 5 and 11 as 4 WITHOUT fences: the payload goes out as agent-scope (write-through) stores, is waited for (vmcnt) and read back
//      with agent-scope loads -- no buffer_wbl2 / buffer_inv
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                           \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

__device__ __forceinline__ unsigned peek(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void poke(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every wait is bounded (20 ms of the 100 MHz clock): a wrong assumption shows as ws[8] != 0, not as a hung box
#define SPIN(cond, sl)                                                     \
    do {                                                                   \
        const long long t0_ = wall_clock64();                              \
        while (!(cond)) {                                                  \
            __builtin_amdgcn_s_sleep(sl);                                  \
            if (wall_clock64() - t0_ > 2000000LL) { poke(ws + 8, 1u); break; } \
        }                                                                  \
    } while (0)
__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v & 15u;
}

// ws layout (uint32): [0] global count, [16] generation, [32 + 16 x] per-XCD count, [256 + 16 x] per-XCD flag, [1024 + 16 w] per-wave flag,
// [1024 + 16 * 2048 + w] per-wave arrival flags (contiguous)
template <int V>
__global__ __launch_bounds__(64) void k_meet(unsigned* ws, unsigned gen, float* sink)
{
    const int w = blockIdx.x, total = gridDim.x, lane = threadIdx.x;
    if (V == 0) return;
    constexpr bool NOFENCE = V == 10 || V == 11;
    if (V == 9) {
        if (lane == 0) __hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // some payload the release has to publish
    if (NOFENCE) {
        __hip_atomic_store(sink + (size_t)w * 64 + lane, (float)gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);       // the store has left the wave ... (vmcnt = 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        sink[(size_t)w * 64 + lane] = (float)gen;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    }
    if (V == 8) return;
    unsigned* wave_flag = ws + 1024 + 16 * w;
    unsigned* arrive = ws + 1024 + 16 * 2048;
    if (V == 1 || V == 2 || V == 3 || V == 6) {
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        if (V == 1) return;
        if (V == 2 || V == 6) {
            SPIN(peek(ws) >= gen * total, V == 6 ? 8 : 1);
        } else {
            if (old == gen * total - 1u)
                for (int i = lane; i < total; i += 64) poke(ws + 1024 + 16 * i, gen);
            SPIN(peek(wave_flag) == gen, 1);
        }
    } else if (V == 4 || V == 7 || V == 11) {
        const unsigned x = xcc_id();
        unsigned* xc = ws + 32 + 16 * x;
        unsigned* xf = ws + 256 + 16 * x;
        // how many waves of this launch sit on XCD x: round robin over 8 XCDs by workgroup id
        const unsigned mine = (total + 7 - x) / 8;
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(xc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == gen * mine - 1u) {       // the XCD's last arrival
            if (lane == 0) __hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            SPIN(peek(ws) >= gen * 8u, 1);
            if (V == 4 || V == 11) {
                if (lane == 0) poke(xf, gen);
            } else {
                for (unsigned i = x + 8 * lane; i < (unsigned)total; i += 8 * 64) poke(ws + 1024 + 16 * i, gen);
            }
        }
        if (V == 4 || V == 11) {
            SPIN(peek(xf) == gen, 1);
        } else {
            SPIN(peek(wave_flag) == gen, 1);
        }
    } else if (V == 5 || V == 10) {
        if (lane == 0) poke(arrive + w, gen);
        if (w == 0) {
            const long long t0 = wall_clock64();
            for (;;) {
                unsigned ok = 1;
                for (int i = lane; i < total; i += 64) ok &= peek(arrive + i) == gen;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 2000000LL) { poke(ws + 8, 1u); break; }
            }
            for (int i = lane; i < total; i += 64) poke(ws + 1024 + 16 * i, gen);
        }
        SPIN(peek(wave_flag) == gen, 1);
    }
    // touch what the others published
    if (NOFENCE) {
        sink[(size_t)total * 64 + (size_t)w * 64 + lane] = __hip_atomic_load(sink + (size_t)((w + 97) % total) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        sink[(size_t)total * 64 + (size_t)w * 64 + lane] = sink[(size_t)((w + 97) % total) * 64 + lane];
    }
}

__global__ void k_census(unsigned* ids)
{
    if (threadIdx.x == 0) ids[blockIdx.x] = xcc_id();
}

template <int V>
float run(int waves, unsigned* ws, float* sink, int reps)
{
    CHECK(hipMemset(ws, 0, (1024 + 16 * 2048 + 4096) * 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    unsigned gen = 0;
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_meet<V>, dim3(waves), dim3(64), 0, 0, ws, ++gen, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_meet<V>, dim3(waves), dim3(64), 0, 0, ws, ++gen, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    unsigned flag = 0;
    CHECK(hipMemcpy(&flag, ws + 8, 4, hipMemcpyDeviceToHost));
    if (flag) printf(" [variant %d: a wait ran into its time limit] ", V);
    return ms * 1e3f / reps;
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    unsigned* ws;
    float* sink;
    CHECK(hipMalloc(&ws, (1024 + 16 * 2048 + 4096) * 4));
    CHECK(hipMalloc(&sink, (size_t)2 * 2048 * 64 * 4));
    // is the round-robin assumption of variant 4 right (workgroup i on XCD i % 8)?
    {
        unsigned* ids;
        CHECK(hipMalloc(&ids, 1024 * 4));
        hipLaunchKernelGGL(k_census, dim3(1024), dim3(64), 0, 0, ids);
        std::vector<unsigned> h(1024);
        CHECK(hipMemcpy(h.data(), ids, 4096, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 1024; ++i) bad += h[i] != (unsigned)(i % 8);
        printf("XCC_ID census: %d of 1024 workgroups NOT on XCD (id %% 8); first 16:", bad);
        for (int i = 0; i < 16; ++i) printf(" %u", h[i]);
        printf("\n");
    }
    for (int waves : {256, 1022, 1024}) {
        printf("waves %4d:", waves);
        printf("  empty %.2f", run<0>(waves, ws, sink, reps));
        printf("  arrive %.2f", run<1>(waves, ws, sink, reps));
        printf("  arrive+poll-one-word %.2f", run<2>(waves, ws, sink, reps));
        printf("  ...sleep8 %.2f", run<6>(waves, ws, sink, reps));
        printf("  per-wave-flags %.2f", run<3>(waves, ws, sink, reps));
        printf("  per-XCD %.2f", run<4>(waves, ws, sink, reps));
        printf("  per-XCD+wave-flags %.2f", run<7>(waves, ws, sink, reps));
        printf("  no-atomics %.2f", run<5>(waves, ws, sink, reps));
        printf("  | fence-only %.2f", run<8>(waves, ws, sink, reps));
        printf("  atomic-only %.2f", run<9>(waves, ws, sink, reps));
        printf("  no-atomics-no-fence %.2f", run<10>(waves, ws, sink, reps));
        printf("  per-XCD-no-fence %.2f", run<11>(waves, ws, sink, reps));
        printf("  us per launch\n");
        fflush(stdout);
    }
    return 0;
}
