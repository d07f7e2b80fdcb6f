// r02 probe: where do the 12 us of one k_env_step launch (65 536 agents, one wave per SIMD) go?
// Same device code as the product (the .hip sources are included), launched in ablated forms, timed with HIP events over
// back-to-back launches.  Built and driven by tools/exp_env_ablate.py.
#include "../visfly_amd/csrc/vf_dyn.hip"
#include "../visfly_amd/csrc/vf_env.hip"

#include <vector>

using namespace vf;

// MODE 0 full step | 1 no stores at all (results folded into one word per wave) | 2 state loads + state stores only (no arithmetic)
//      3 loads only | 4 empty kernel | 5 full arithmetic on register-resident constants (no loads), stores as usual
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_probe(const vf_dyn_cfg c, const vf_env_cfg e, const EnvArgs g, float* sink)
{
    __shared__ __attribute__((aligned(16))) float tile[kBlock * 13];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if constexpr (MODE == 4) {
        if (g.d.N < 0) sink[i] = 1.0f;
        return;
    }
    const bool live = i < g.d.N;
    Agent s;
    Spares sp;
    float a[4], head_bits = 0.0f;
    if constexpr (MODE == 5) {
        s.t = 0.f;
        for (int k = 0; k < 3; ++k) { s.p[k] = 1.0f + 0.001f * (i & 7) + k; s.v[k] = 0.01f * k; s.w[k] = 0.02f * k; s.aa[k] = 0.f; s.acc[k] = 0.f; }
        s.q = Quat{1.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < 4; ++k) { s.wm[k] = c.w_init; s.T[k] = c.T_init; a[k] = k == 0 ? -0.33f : 0.01f * (i & 3); }
        sp = Spares{0.f, 0.f, 0.f, 0.f};
    } else {
        ring_exchange(c, g.d, i, live, head_bits, a);
        load_agent<false>(g.d.S, g.d.G, i, s, sp);
        if (c.delay_steps > 0) sp.vel = head_bits;
    }
    if constexpr (MODE == 3) {
        float acc = s.t + s.p[0] + s.q.w + s.v[0] + s.w[0] + s.wm[0] + s.aa[0] + sp.acc + a[0];
        if (acc == 123.456f) sink[i] = acc;
        return;
    }
    if constexpr (MODE == 2) {
        store_agent(g.d.S, g.d.G, i, s, sp);
        return;
    }
    float kl[3], kq[3];
    drag_of(c, g.d, i, kl, kq);
    control_interval<VF_ACT_BODYRATE, VF_INT_EULER, true>(c, s, a, kl, kq);
    const int wave = threadIdx.x >> 6;
    if constexpr (MODE == 1) {
        const Collision col = bbox_collision(e, s.p);
        const float vel[3] = {s.v[0], s.v[1], s.v[2]};
        float r = hover_reward(s.p, e.target, s.q, vel, s.w) + col.dis + s.T[0] + s.acc[1] + s.aa[2] + s.wm[3] + s.t;
        if (r == 123.456f) sink[i] = r;
        return;
    }
    env_epilogue<VF_ENV_HOVER>(c, e, g, i, live, s, sp, blockIdx.x * kBlock + wave * 64, tile + wave * 64 * 13);
}

// exhaustive checks of the VF_FAST_EXACT fast paths against the compiler's IEEE expansions
__global__ void k_check_sqrt(unsigned long long* bad, unsigned* first)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b <= 0x7f800000ull; b += stride) {
        const float x = __uint_as_float((unsigned)b);
        if (!(x == 0.0f || x >= 0x1p-96f)) continue;
        const float a = sqrt_exact_core(x), r = sqrtf(x);
        if (__float_as_uint(a) != __float_as_uint(r)) { if (atomicAdd(bad, 1ull) == 0) *first = (unsigned)b; }
    }
}
__global__ void k_check_div(float m, unsigned long long* bad, unsigned* first)
{
    const float y = 1.0f / m;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((unsigned)b);
        const float ax = __builtin_fabsf(x);
        if (!(ax >= 0x1p-100f && ax <= 0x1p100f)) continue;
        const float a = div_const_core(x, m, y), r = x / m;
        if (__float_as_uint(a) != __float_as_uint(r)) { if (atomicAdd(bad, 1ull) == 0) *first = (unsigned)b; }
    }
}

void exhaustive_checks(float mass)
{
    unsigned long long* bad;
    unsigned* first;
    hipMalloc(&bad, 8);
    hipMalloc(&first, 4);
    auto report = [&](const char* what) {
        unsigned long long hb = 0;
        unsigned hf = 0;
        hipDeviceSynchronize();
        hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
        printf("exhaustive %-60s mismatches %llu%s\n", what, hb, hb ? " (first bits below)" : "");
        if (hb) printf("    first mismatching bit pattern 0x%08x\n", hf);
    };
    hipMemset(bad, 0, 8);
    k_check_sqrt<<<4096, 256>>>(bad, first);
    report("sqrt_exact_core vs sqrtf, x = 0 and all x >= 2^-96:");
    for (float m : {mass, 0.46f, 0.68f, 1.0f, 1.5f, 0.033f, 3.3f, 0.7500001f}) {
        hipMemset(bad, 0, 8);
        k_check_div<<<4096, 256>>>(m, bad, first);
        char buf[96];
        snprintf(buf, sizeof buf, "div_const_core vs x / m, m = %.9g, 2^-100 <= |x| <= 2^100:", m);
        report(buf);
    }
    hipFree(bad);
    hipFree(first);
}

// the full step with the two constant blocks read through pointers into persistent device memory instead of by-value kernel
// arguments (a fresh kernarg block per launch)
__global__ __launch_bounds__(kBlock) void k_probe_ptr(const vf_dyn_cfg* cp, const vf_env_cfg* ep, const EnvArgs g)
{
    __shared__ __attribute__((aligned(16))) float tile[kBlock * 13];
    const vf_dyn_cfg& c = *cp;
    const vf_env_cfg& e = *ep;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < g.d.N;
    Agent s;
    Spares sp;
    float a[4], head_bits = 0.0f;
    ring_exchange(c, g.d, i, live, head_bits, a);
    load_agent<false>(g.d.S, g.d.G, i, s, sp);
    if (c.delay_steps > 0) sp.vel = head_bits;
    float kl[3], kq[3];
    drag_of(c, g.d, i, kl, kq);
    control_interval<VF_ACT_BODYRATE, VF_INT_EULER, true>(c, s, a, kl, kq);
    const int wave = threadIdx.x >> 6;
    env_epilogue<VF_ENV_HOVER>(c, e, g, i, live, s, sp, blockIdx.x * kBlock + wave * 64, tile + wave * 64 * 13);
}

float time_ptr(const vf_dyn_cfg& c, const vf_env_cfg& e, const EnvArgs& g, int blocks, int iters)
{
    vf_dyn_cfg* cp;
    vf_env_cfg* ep;
    hipMalloc(&cp, sizeof(c));
    hipMalloc(&ep, sizeof(e));
    hipMemcpy(cp, &c, sizeof(c), hipMemcpyHostToDevice);
    hipMemcpy(ep, &e, sizeof(e), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(k_probe_ptr, dim3(blocks), dim3(kBlock), 0, 0, cp, ep, g);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int k = 0; k < iters; ++k) hipLaunchKernelGGL(k_probe_ptr, dim3(blocks), dim3(kBlock), 0, 0, cp, ep, g);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(cp);
    hipFree(ep);
    return ms * 1000.0f / iters;
}

template <int MODE>
float time_mode(const vf_dyn_cfg& c, const vf_env_cfg& e, const EnvArgs& g, float* sink, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(kBlock), 0, 0, c, e, g, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int k = 0; k < iters; ++k) hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(kBlock), 0, 0, c, e, g, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / iters;
}

int main(int argc, char** argv)
{
    vf_dyn_cfg c;
    vf_env_cfg e;
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(&c, sizeof(c), 1, f) != 1 || fread(&e, sizeof(e), 1, f) != 1) { printf("cfg file?\n"); return 1; }
    fclose(f);
    printf("VF_FAST_EXACT=%d VF_STORE_MODE=%d\n", VF_FAST_EXACT, VF_STORE_MODE);
    if (argc > 2) exhaustive_checks(c.m);
    for (int N : {64, 32768, 65536, 131072, 1048576}) {
        const int G = VF_G_FIXED + c.delay_steps;
        const int blocks = (N + kBlock - 1) / kBlock;
        const size_t Npad = (size_t)blocks * kBlock;
        float *S, *act, *obs, *rew, *sink, *epr, *tobs;
        uint8_t *done, *epf;
        int32_t* epl;
        hipMalloc(&S, Npad * G * 16); hipMalloc(&act, Npad * 16); hipMalloc(&obs, Npad * 52); hipMalloc(&rew, Npad * 4);
        hipMalloc(&sink, Npad * 4); hipMalloc(&done, Npad); hipMalloc(&epr, Npad * 4); hipMalloc(&epl, Npad * 4);
        hipMalloc(&epf, Npad); hipMalloc(&tobs, Npad * 52);
        std::vector<float> h(Npad * G * 4, 0.f);
        for (size_t i = 0; i < Npad; ++i) {
            auto at = [&](int g, int k) -> float& { return h[(((i >> 6) * G + g) * 64 + (i & 63)) * 4 + k]; };
            at(0, 1) = 1.f + 0.001f * (i % 97); at(0, 3) = 1.5f; at(1, 0) = 1.f;
            for (int k = 0; k < 4; ++k) { at(4, k) = c.w_init; at(5, k) = c.T_init; }
        }
        hipMemcpy(S, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> ha(Npad * 4);
        for (size_t i = 0; i < Npad; ++i) { ha[4 * i] = -0.3333f + 0.001f * (i % 7); ha[4 * i + 1] = 0.01f; ha[4 * i + 2] = -0.01f; ha[4 * i + 3] = 0.f; }
        hipMemcpy(act, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
        EnvArgs g{DynArgs{N, G, -1, S, reinterpret_cast<const float4*>(act), obs, 0},
                  vf_env_out{obs, rew, done, epr, epl, epf, tobs, nullptr, nullptr, nullptr}, -1, 1};
        const int iters = N > 200000 ? 100 : 400;
        printf("N=%8d:", N);
        for (int sub : {8, 0, 1, 4, 16}) {
            vf_dyn_cfg cc = c;
            cc.interval_steps = sub;
            printf("  full(sub=%d) %.2f", sub, time_mode<0>(cc, e, g, sink, blocks, iters));
        }
        printf("  | cfg through device pointers %.2f", time_ptr(c, e, g, blocks, iters));
        printf("\n           no-stores %.2f | state load+store only %.2f | loads only %.2f | empty %.2f | no-loads(+stores) %.2f us\n",
               time_mode<1>(c, e, g, sink, blocks, iters), time_mode<2>(c, e, g, sink, blocks, iters),
               time_mode<3>(c, e, g, sink, blocks, iters), time_mode<4>(c, e, g, sink, blocks, iters),
               time_mode<5>(c, e, g, sink, blocks, iters));
        hipFree(S); hipFree(act); hipFree(obs); hipFree(rew); hipFree(sink); hipFree(done); hipFree(epr); hipFree(epl); hipFree(epf); hipFree(tobs);
    }
    return 0;
}
