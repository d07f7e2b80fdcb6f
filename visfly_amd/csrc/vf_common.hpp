// Shared host-side plumbing for libvisfly_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "visfly_amd.h"

namespace vf {

char* err_buf();  // thread-local message buffer behind vf_last_error()

inline int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#ifdef VF_CHAIN_PLUGIN
// a chain plugin (vf_chain_plugin.hpp) is a shared object of its own: it calls nothing of libvisfly_amd.so, the library words the error
#define VF_HIP(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return -1000 - (int)e_;                                       \
    } while (0)
#else
#define VF_HIP(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return vf::fail(VF_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                            \
    } while (0)
#endif

inline hipStream_t as_stream(vf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kBlock = 256;  // 4 wave64 per workgroup

inline int blocks_for(int n) { return (n + kBlock - 1) / kBlock; }

// Kernel shape per launch: while one-thread-per-agent work cannot even give every SIMD a wave
// (<= 32768 agents) the two-wave split (rotation | translation on co-resident waves) wins; from one
// wave per SIMD on, the plain kernel issues fewer instructions in total and has no hand-off barriers.  VISFLY_AMD_SPLIT=0/1 forces a shape (A/B experiments).
inline bool use_split(int agents_padded, const vf_dyn_cfg& cfg)
{
    // the geometric controller (velocity / position) needs the whole state in one thread
    if (cfg.action_type != VF_ACT_THRUST && cfg.action_type != VF_ACT_BODYRATE) return false;
    static const int forced = [] {
        const char* e = getenv("VISFLY_AMD_SPLIT");
        return e ? atoi(e) : -1;
    }();
    if (forced >= 0) return forced != 0;
    return agents_padded <= 32768;   // measured on MI355X: 32768 agents 9.4 us (split) vs 11.3 us; 65536: 13.3 vs 12.5
}

#ifdef __HIPCC__
// ---- Philox4x32-10 counter RNG (on-device spawner, exploration noise of the policy head) ----
struct U4 {
    unsigned x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(U4 ctr, unsigned k0, unsigned k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
        ctr = U4{hi1 ^ ctr.y ^ k0, lo1, hi0 ^ ctr.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return ctr;
}

__device__ __forceinline__ float u01(unsigned x) { return (float)(x & 0xFFFFFFu) * (1.0f / 16777216.0f); }

// ---- activations of the MLP layers (vf_mlp_layer.relu / vf_mlp_bwd_layer.act: VF_ACTIVATION_*) ----
// create_mlp's `activation_fn` (utils/policies/extractors.py:376-449; the aliases of policies.py:64-69): ReLU, Tanh, ELU (alpha = 1),
// LeakyReLU (slope 0.01) -- torch.nn defaults.  The derivative is a function of the OUTPUT in all four, so the reverse sweep needs
// the saved layer output only (as for the ReLU mask): d/dz = [y > 0] | 1 - y^2 | y > 0 ? 1 : y + 1 | y > 0 ? 1 : 0.01
__device__ __forceinline__ float act_fwd(float z, int kind)
{
    switch (kind) {
    case VF_ACTIVATION_RELU: return z > 0.0f ? z : 0.0f;
    case VF_ACTIVATION_TANH: return tanhf(z);
    case VF_ACTIVATION_ELU: return z > 0.0f ? z : expm1f(z);
    case VF_ACTIVATION_LEAKY_RELU: return z > 0.0f ? z : 0.01f * z;
    default: return z;
    }
}
// upstream gradient v through the activation whose output was y
__device__ __forceinline__ float act_mul(float v, float y, int kind)
{
    switch (kind) {
    case VF_ACTIVATION_TANH: return v * (1.0f - y * y);
    case VF_ACTIVATION_ELU: return y > 0.0f ? v : v * (y + 1.0f);
    case VF_ACTIVATION_LEAKY_RELU: return y > 0.0f ? v : 0.01f * v;
    default: return y > 0.0f ? v : 0.0f;        // ReLU (kind 0 with a saved output = ReLU: tables written before ABI 10)
    }
}

// the same with the kind a compile-time constant (the register-chained kernels: ChainLayer::relu); ReLU keeps the max instruction the
// built-in classes have always used
template <int KIND>
__device__ __forceinline__ float act_fwd_c(float z)
{
    if constexpr (KIND == VF_ACTIVATION_RELU) return fmaxf(z, 0.0f);
    else return act_fwd(z, KIND);
}
template <int KIND>
__device__ __forceinline__ float act_mul_c(float v, float y)
{
    if constexpr (KIND <= VF_ACTIVATION_RELU) return y > 0.0f ? v : 0.0f;
    else return act_mul(v, y, KIND);
}

// Pull the whole kernel-argument block into the scalar cache with one batch of loads (24 lines per batch).  The chain kernels
// take their layer tables by value (1.2 - 2.8 KB of kernel arguments at a fresh address every launch) and the compiler fetches a
// field where it is first used: k_ppo_update_chain had 314 s_load / 216 s_waitcnt lgkmcnt in its body, ~44 of them first touches
// of a 64-byte line that go all the way to memory, and with ONE wave per SIMD nothing hides such a stall.  One dword per line,
// kept alive by an empty asm that wants it in an SGPR; afterwards every field load hits the scalar cache.
template <int BYTES>
__device__ __forceinline__ void prefetch_kernarg()
{
    typedef const unsigned __attribute__((address_space(4))) * kptr;
    const kptr w = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int L = (BYTES + 63) / 64;
#pragma unroll
    for (int b0 = 0; b0 < L; b0 += 24) {
        unsigned x[24];
#pragma unroll
        for (int k = 0; k < 24; ++k) x[k] = b0 + k < L ? w[16 * (b0 + k)] : 0u;
        asm volatile("" ::"s"(x[0]), "s"(x[1]), "s"(x[2]), "s"(x[3]), "s"(x[4]), "s"(x[5]), "s"(x[6]), "s"(x[7]), "s"(x[8]), "s"(x[9]),
                     "s"(x[10]), "s"(x[11]), "s"(x[12]), "s"(x[13]), "s"(x[14]), "s"(x[15]), "s"(x[16]), "s"(x[17]), "s"(x[18]),
                     "s"(x[19]), "s"(x[20]), "s"(x[21]), "s"(x[22]), "s"(x[23]));
    }
}

// The same for a constant block behind a kernel-argument POINTER (the persistent device copies of vf_dyn_cfg / vf_env_cfg the step kernels
// read their constants through): one dword of each of the first LINES_A / LINES_B 64-byte lines of two blocks (+ one more word), all
// requested in one batch and waited for once.
// Without it the compiler fetches a field where it is first used, and every first touch of a line is a scalar-cache miss that a lone wave
// sits out in full (L2 round trip): the env step touched ~12 lines one after the other between its loads and its stores.
template <int LINES_A, int LINES_B>
__device__ __forceinline__ void prefetch_const_lines(const void* __restrict__ a, const void* __restrict__ b, const void* __restrict__ c1)
{
    static_assert(LINES_A + LINES_B + 1 <= 16, "one batch");
    const unsigned* wa = reinterpret_cast<const unsigned*>(a);
    const unsigned* wb = reinterpret_cast<const unsigned*>(b);
    unsigned x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
        x[k] = k < LINES_A ? wa[16 * k] : k < LINES_A + LINES_B ? wb[16 * (k - LINES_A)] : k == LINES_A + LINES_B ? *reinterpret_cast<const unsigned*>(c1) : 0u;
    asm volatile("" ::"s"(x[0]), "s"(x[1]), "s"(x[2]), "s"(x[3]), "s"(x[4]), "s"(x[5]), "s"(x[6]), "s"(x[7]), "s"(x[8]), "s"(x[9]),
                 "s"(x[10]), "s"(x[11]), "s"(x[12]), "s"(x[13]), "s"(x[14]), "s"(x[15]));
}

// Both in ONE batch, for a kernel whose pointers to the constant blocks are preloaded kernel arguments (k_env_step): the remaining
// kernel-argument lines do not have to arrive before the constant blocks can be asked for
template <int KBYTES, int LINES_A, int LINES_B>
__device__ __forceinline__ void prefetch_kernarg_and_const_lines(const void* __restrict__ a, const void* __restrict__ b, const void* __restrict__ c1)
{
    typedef const unsigned __attribute__((address_space(4))) * kptr;
    const kptr wk = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int LK = (KBYTES + 63) / 64;
    static_assert(LK + LINES_A + LINES_B + 1 <= 24, "one batch");
    const unsigned* wa = reinterpret_cast<const unsigned*>(a);
    const unsigned* wb = reinterpret_cast<const unsigned*>(b);
    unsigned x[24];
#pragma unroll
    for (int k = 0; k < 24; ++k)
        x[k] = k < LK ? wk[16 * k] : k < LK + LINES_A ? wa[16 * (k - LK)] : k < LK + LINES_A + LINES_B ? wb[16 * (k - LK - LINES_A)]
             : k == LK + LINES_A + LINES_B ? *reinterpret_cast<const unsigned*>(c1) : 0u;
    asm volatile("" ::"s"(x[0]), "s"(x[1]), "s"(x[2]), "s"(x[3]), "s"(x[4]), "s"(x[5]), "s"(x[6]), "s"(x[7]), "s"(x[8]), "s"(x[9]),
                 "s"(x[10]), "s"(x[11]), "s"(x[12]), "s"(x[13]), "s"(x[14]), "s"(x[15]), "s"(x[16]), "s"(x[17]), "s"(x[18]),
                 "s"(x[19]), "s"(x[20]), "s"(x[21]), "s"(x[22]), "s"(x[23]));
}
#endif

// action head fused into the chain kernels (vf_mlp_forward_act / vf_mlp_backward_data_act); all-null: off
struct ReparamFwd {
    const float* log_std;
    const float* eps;
    float* action;
    float* obs_copy[2];
};
struct ReparamBwd {
    const float* d_action;
    const float* action;
    const float* log_std;
    const float* eps;
    float* g_log_std;
};
// vf_mlp_chain.hip: register-chained forward for the reference-default network shapes (1: launched, 0: no match)
int mlp_forward_chain_try(const vf_mlp_desc* d, const float* params, const float* packed, const float* in0, const float* in1,
                          float* out0, float* out1, int M, hipStream_t st, const ReparamFwd* rp = nullptr, const float* in2 = nullptr,
                          int M_choice = 0);      // M_choice > 0: rows-per-wave choice as for M_choice rows (vf_mlp_forward_steps)

int bwd_chain_policy_class(const vf_mlp_bwd_desc* d, int M);           // vf_mlp_chain.hip: 0 none, 1 NetHover, 2 NetNav (policy trunk, obs gradient), 3 / 4 NetSacHover / NetSacNav (both trunks); + 16: M rows per pass run 16 rows per wave
int chain16_policy_class(const vf_mlp_desc* d, const float* params);   // vf_mlp_chain.hip: 0 none, 1 NetHoverPi, 2 NetNavPi, 3 NetSacHover, 4 NetSacNav
int chain_full_class(const vf_mlp_desc* d, const float* params, int M);   // 0 none, 1 NetHover, 2 NetNav (both trunks); + 16: M rows run on the 16-row chain
// reverse chain (data gradients) for the same network classes: 1 launched, 0 no match, < 0 error
int mlp_backward_chain_try(const vf_mlp_bwd_desc* d, const float* packed, int M, hipStream_t st, const ReparamBwd* rp = nullptr);
// the same for the SAC-style Actor's network classes (vf_mlp_chain_sac.hip); called by the two functions above as their last resort
int mlp_forward_chain_try_sac(const vf_mlp_desc* d, const float* params, const float* packed, const float* in0, const float* in1,
                              const float* in2, float* out0, float* out1, int M, hipStream_t st, int M_choice = 0);
int mlp_backward_chain_try_sac(const vf_mlp_bwd_desc* d, const float* packed, int M, hipStream_t st);
// vf_mlp_chain_sac.hip: fused critic step of SHAC (forward + twin-Q loss + reverse chain) for the ContinuousCritic class: 1 launched, 0 no match
int twin_q_update_chain_try(const vf_mlp_desc* d, const vf_mlp_bwd_desc* bd, const float* params, const float* packed, const float* in0,
                            const float* in1, const float* target, double* part, float scale, int M, hipStream_t st);
int ppo_update_chain_try(const vf_mlp_desc* d, const vf_mlp_bwd_desc* bd, const float* params, const float* packed, const float* in0,
                         const float* in1, const float* log_std, const float* action, const float* old_lp, const float* adv,
                         const float* ret, float* part, const vf_ppo_loss_cfg* cfg, int M, hipStream_t st);
// vf_mlp_wgrad.hip: weight / bias gradients of every listed layer from dY (masked) and X, + fold into grad
int64_t mlp_wgrad_partial_floats(const vf_mlp_bwd_desc* d, int M);
int mlp_wgrad_fold_blocks(const vf_mlp_bwd_desc* d);
int mlp_wgrad_launch(const vf_mlp_bwd_desc* d, float* partials, float* grad, int M, int accumulate, double* sq_part,
                     const vf_stats_fold* loss_stats, hipStream_t st);
int mlp_wgrad_launch_layers(const vf_mlp_bwd_desc* d, float* partials, float* grad, int M, int accumulate, unsigned layer_mask, hipStream_t st);
// the same with fold, gradient norm, clip and Adam inside the weight-gradient launch: 1 launched, 0 not for this table / device, < 0 error
int mlp_wgrad_adam_launch(const vf_mlp_bwd_desc* d, float* partials, float* grad, int M, int accumulate, const vf_stats_fold* loss_stats,
                          const vf_wgrad_tail* tail, hipStream_t st);

}  // namespace vf
