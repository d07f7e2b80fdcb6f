// vf_mlp_chain_split.hip -- the fused update kernels (forward chain + loss + reverse chain of a minibatch in one launch) in their
// branch-parallel form: a workgroup of two waves per tile of 32 rows, each wave walks half of the network (vf_mlp_chain_split.hpp).
//   k_ppo_update_split    PPO minibatch step (PPO.py:210-263) for the actor-critic classes of policies.py:18-49
//   k_twin_q_update_split SHAC critic step (shac.py:267-270) for the twin ContinuousCritic of td_policies.py:82-143
// Same arguments, same buffers left behind for k_mlp_wgrad as their one-wave forms (vf_mlp_chain.hip, vf_mlp_chain_sac.hip), which stay
// the fallback (VISFLY_AMD_CHAIN_SPLIT=0 forces them: A/B).
#ifdef VF_SPLIT_TRACE
#include <hip/hip_runtime.h>
// [tile][role][32]: 0 HW_ID | XCC_ID << 32, 1 realtime at exit, 2 realtime at entry, 3 .. 7 phase stamps, 8 + 5 kind + idx: layer / op stamps
__device__ unsigned long long vf_split_trace[2 * 8192][32];
#define VF_CHAIN_HOOK(kind, idx) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 8192) vf_split_trace[2 * blockIdx.x + (threadIdx.x >> 6)][8 + 5 * (kind) + (idx)] = __builtin_readcyclecounter(); } while (0)
#endif
#include "vf_mlp_chain_kernels.hpp"
#include "vf_mlp_chain_split.hpp"

namespace vf {

__shared__ float vf_xch_q[2][64];     // the twin critic's two heads meet for min(Q1, Q2): [role][lane]

#ifdef VF_SPLIT_TRACE
// placement + timeline of every wave of the last k_ppo_update_split launch (layout: at the head of the file)
#define VF_STRACE(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 8192) vf_split_trace[2 * blockIdx.x + R][k] = __builtin_readcyclecounter(); } while (0)
#else
#define VF_STRACE(k) do { } while (0)
#endif

// Which form runs M rows?  Two half-chain waves per tile finish a tile in about 0.6 of the one-wave time while every wave still has a SIMD
// to itself (34.0 vs 55.1 us at 8 192 rows, 53.7 vs 56.6 at 16 384); with two waves per SIMD most of that goes to the second prologue,
// the hand-over barriers and a lower clock (25 600 rows 59.3 vs 60.5, 32 768 rows 62.3 vs 65.5 us under rocprofv3), and over the
// 524 288 rows of a SHAC critic update -- many tiles per SIMD either way -- the one-wave form is 2 % ahead: profiles/r05_chain_split.txt.
// VISFLY_AMD_CHAIN_SPLIT=0/1 forces a form (A/B, tests); read per call.
static bool chain_split_for(int M)
{
    const char* e = getenv("VISFLY_AMD_CHAIN_SPLIT");
    if (e && *e) return atoi(e) != 0;
    return M <= 32768;
}

template <class N, int R>
__device__ __forceinline__ void ppo_update_role(const ChainArgs& g, const BwdArgsChain& gb, const PpoRowArgs& pr)
{
    using S = SplitNet<N, R>;
    using P = SplitBwd<N, R>;
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int row = blockIdx.x * 32 + m;
    const bool live = row < g.M;
    const int rc = live ? row : g.M - 1;
#ifdef VF_SPLIT_TRACE
    if (lane == 0 && blockIdx.x < 8192) {
        vf_split_trace[2 * blockIdx.x + R][0] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                                 ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF) << 32);
        vf_split_trace[2 * blockIdx.x + R][2] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    VF_STRACE(3);
    // per-row loss inputs first: the action-only part of the loss (ppo_row_pre) runs while the weight fragments are on their way
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float old_lp = 0.0f, adv = 0.0f, ret = 0.0f, ls[4] = {0.f, 0.f, 0.f, 0.f};
    const int rs = ppo_source_row(pr, rc);
    if constexpr (R == 0) {
        a4 = pr.action[rs];
#pragma unroll
        for (int k = 0; k < 4; ++k) ls[k] = pr.log_std[k];
        old_lp = pr.old_lp[rs];
        adv = pr.adv[rc];
    } else {
        ret = pr.ret[rs];
    }
    ChainState<S> fs;
    chain_prologue<S, 0>(g, fs, lane);
    split_load_obs<N, R>(g, fs, rs, h, rc, live);
    PpoRowPre pre{};
    if constexpr (R == 0) {
        const float a[4] = {a4.x, a4.y, a4.z, a4.w};
        pre = ppo_row_pre(a);
#pragma unroll
        for (int d = 0; d < 4; ++d)     // pinned here (k_ppo_update_chain)
            asm volatile("" : "+v"(pre.g[d]), "+v"(pre.corr[d]));
    }
    chain_items<S, 0>(g, fs, lane, row, live, rc);
    VF_STRACE(4);
    BwdState<P> bs;
    bwd_prologue<P, 0>(gb, bs, lane);                  // first weight blocks of the reverse chain: in flight during the loss arithmetic
    // ---- this role's part of the row's loss (ppo_row's arithmetic; what the role does not use is dead code) ----
    const bool on = live && h == 0;
    if constexpr (R == 0) {
        const f32x16& mt = fs.t[N::t_mean];
        const float mu[4] = {mt[0], mt[1], mt[2], mt[3]};
        float st1[9], dm1[4], dv1;
        ppo_row_post(pre, mu, 0.0f, ls, old_lp, adv, 0.0f, pr.cfg, dm1, dv1, st1, rc);
        // head gradients in lane half 0 of EVERY lane: lanes past the last row are replicas of row M - 1 and stay replicas through the
        // reverse chain, whose dZ stores are unguarded (k_ppo_update_chain)
#pragma unroll
        for (int k = 0; k < 4; ++k) bs.hin[0][k] = h == 0 ? dm1[k] : 0.0f;
        if (on) {
            const vf_mlp_bwd_layer& Em = gb.d.layer[P::entry(P::L_mean)];
            *reinterpret_cast<float4*>(const_cast<float*>(Em.dY) + (size_t)row * Em.ld_dy) = make_float4(dm1[0], dm1[1], dm1[2], dm1[3]);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (k == 1) continue;
            float s = on ? st1[k] : 0.0f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
            if (lane == 0) pr.part[(size_t)blockIdx.x * kStats + k] = s;
        }
    } else {
        const float mu[4] = {0.f, 0.f, 0.f, 0.f};
        float st1[9], dm1[4], dv1;
        ppo_row_post(pre, mu, fs.t[N::t_val][0], ls, 0.0f, 0.0f, ret, pr.cfg, dm1, dv1, st1, rs);     // (only the value terms are used)
        const float dvl = h == 0 ? dv1 : 0.0f;
        bs.hin[1][0] = dvl; bs.hin[1][1] = 0.0f; bs.hin[1][2] = 0.0f; bs.hin[1][3] = 0.0f;
        if (on) {
            const vf_mlp_bwd_layer& Ev = gb.d.layer[P::entry(P::L_val)];
            const_cast<float*>(Ev.dY)[(size_t)row * Ev.ld_dy] = dvl;
        }
        float s = on ? st1[1] : 0.0f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) pr.part[(size_t)blockIdx.x * kStats + 1] = s;
    }
    VF_STRACE(5);
    bwd_items<P, ChainState<S>, 0>(gb, bs, fs, lane, row, rc, live);
    VF_STRACE(6);
    bwd_tail_store<P>(gb, bs, row, h, live);
    VF_STRACE(7);
#ifdef VF_SPLIT_TRACE
    if (lane == 0 && blockIdx.x < 8192) vf_split_trace[2 * blockIdx.x + R][1] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <class N>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ppo_update_split(const ChainArgs g, const BwdArgsChain gb, const PpoRowArgs pr)
{
    prefetch_kernarg<sizeof(ChainArgs) + sizeof(BwdArgsChain) + sizeof(PpoRowArgs)>();
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (role == 0) ppo_update_role<N, 0>(g, gb, pr);
    else ppo_update_role<N, 1>(g, gb, pr);
}

// 1 launched, 0 not taken (class, switch), < 0 error.  Called by ppo_update_chain_try after ITS checks of the tables (row counts, saved
// copies); part: ceil(M / 32) x kStats floats
int ppo_update_split_try(const ChainArgs& g, const BwdArgsChain& gb, const void* prv, int which, int M, hipStream_t st)
{
    if (!chain_split_for(M)) return 0;
    const PpoRowArgs& pr = *static_cast<const PpoRowArgs*>(prv);
    const dim3 grid((M + 31) / 32);
    if (which == 2) hipLaunchKernelGGL(k_ppo_update_split<NetNav>, grid, dim3(128), 0, st, g, gb, pr);
    else if (which == 1) hipLaunchKernelGGL(k_ppo_update_split<NetHover>, grid, dim3(128), 0, st, g, gb, pr);
    else return 0;
    VF_HIP(hipGetLastError());
    return 1;
}

// ------------------------------------------------------------------------------------------------
template <class N, int R>
__device__ __forceinline__ void twin_q_update_role(const ChainArgs& g, const BwdArgsChain& gb, const float* __restrict__ target,
                                                   double* __restrict__ part, float scale)
{
    using S = SplitNet<N, R>;
    using P = SplitBwd<N, R>;
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const int row = blockIdx.x * 32 + m;
    const bool live = row < g.M;
    const int rc = live ? row : g.M - 1;
    ChainState<S> fs;
    chain_prologue<S, 0>(g, fs, lane);
    split_load_obs<N, R>(g, fs, rc, h);
    chain_pass_tile<S, R == 0>(g, fs, row, rc, h, live);
    const float tgt = target[rc];
    chain_items<S, 0>(g, fs, lane, row, live, rc);
    BwdState<P> bs;
    bwd_prologue<P, 0>(gb, bs, lane);
    // ---- min(Q1, Q2): the two heads live in different waves; one float per lane through LDS ----
    const float q_own = fs.t[S::t_head][0];
    vf_xch_q[R][lane] = q_own;
    xch_barrier();
    const float q_other = vf_xch_q[1 - R][lane];
    const float q0 = R == 0 ? q_own : q_other, q1 = R == 0 ? q_other : q_own;
    const bool first = q0 <= q1;                       // ties: the first, like torch.min over dim 1
    const float diff = (first ? q0 : q1) - tgt;
    const float gq = 2.0f * diff * scale;
    const bool mine = R == 0 ? first : !first;
    const float dq = (h == 0 && mine) ? gq : 0.0f;
    if (live && h == 0) {
        const vf_mlp_bwd_layer& E = gb.d.layer[P::entry(P::L_head)];
        const_cast<float*>(E.dY)[(size_t)row * E.ld_dy] = dq;
    }
    if constexpr (R == 0) {
        double sq = (live && h == 0) ? (double)diff * (double)diff : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
        if (lane == 0) part[blockIdx.x] = sq;
    }
    bs.hin[R][0] = dq; bs.hin[R][1] = 0.0f; bs.hin[R][2] = 0.0f; bs.hin[R][3] = 0.0f;
    bwd_items<P, ChainState<S>, 0>(gb, bs, fs, lane, row, rc, live);
    bwd_tail_store<P>(gb, bs, row, h, live);
}

template <class N>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_twin_q_update_split(const ChainArgs g, const BwdArgsChain gb, const float* __restrict__ target,
                                                             double* __restrict__ part, float scale)
{
    prefetch_kernarg<sizeof(ChainArgs) + sizeof(BwdArgsChain) + 24>();
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (role == 0) twin_q_update_role<N, 0>(g, gb, target, part, scale);
    else twin_q_update_role<N, 1>(g, gb, target, part, scale);
}

int twin_q_update_split_try(const ChainArgs& g, const BwdArgsChain& gb, const float* target, double* part, float scale, int M, hipStream_t st)
{
    if (!chain_split_for(M)) return 0;
    hipLaunchKernelGGL(k_twin_q_update_split<NetCriticHover>, dim3((M + 31) / 32), dim3(128), 0, st, g, gb, target, part, scale);
    VF_HIP(hipGetLastError());
    return 1;
}

}  // namespace vf

#ifdef VF_SPLIT_TRACE
extern "C" int vf_debug_split_trace(unsigned long long* out, int n_waves)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(vf_split_trace), sizeof(unsigned long long) * 32 * (size_t)n_waves, 0, hipMemcpyDeviceToHost);
}
#endif
