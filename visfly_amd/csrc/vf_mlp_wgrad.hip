// Weight / bias gradients of all layers in one launch, MFMA operands straight from global memory.
//
// dW[n][k] = sum_m dZ[m][n] X[m][k] reduces over rows, so with v_mfma_f32_32x32x2_f32
//     A[i = n][kk] = dZ[m0 + 2 s + kk][n]    lane = n + 32 kk      (row-major dZ: 32 lanes read 128 contiguous bytes)
//     B[kk][j = k] = X [m0 + 2 s + kk][k]    lane = k + 32 kk      (row-major X:  likewise)
// both fragments are plain coalesced dword loads of the buffers the forward / reverse chain kernels left in HBM -- no
// LDS staging, no transposes.  A wave owns ONE layer and a slab of rows: the whole dW of the layer (up to 4 x 4 tiles
// = 256 accumulator registers) stays in registers over the slab, the bias gradient is the running sum of the A
// fragments.  Slabs are sized so that every wave issues about the same number of MFMAs (rows per wave inversely
// proportional to the layer's tile count); each wave writes one partial [No][K] + [No], and a table-driven fold sums
// the partials of a layer in a fixed order (deterministic).
#include "vf_common.hpp"

namespace vf {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// row pairs in flight per wave.  The kernel streams 2 rows x (No + K) floats per step and is bandwidth-bound: by
// Little's law the chip needs ~16 KiB in flight per wave (1024 waves x 16 KiB / ~2.5 us = 6.5 TB/s), i.e.
// 64 / (NT + KT) steps of 256 (NT + KT) bytes
#ifndef VF_WGRAD_BUDGET
#define VF_WGRAD_BUDGET 64       // operand registers of the prefetch ring (A/B knob: profiles/r04_ppo_wgrad.txt)
#endif
#ifndef VF_WGRAD_BUDGET_SMALL
#define VF_WGRAD_BUDGET_SMALL 40
#endif
constexpr int wg_depth(int nt, int kt, bool small)
{
    const int d = (small ? VF_WGRAD_BUDGET_SMALL : VF_WGRAD_BUDGET) / (nt + kt);      // small: two waves per SIMD share its 512 registers
    return d > 16 ? 16 : (d < 4 ? 4 : d);
}

struct WgradTable {
    int32_t n_layers;
    int32_t first_wave[VF_MLP_MAX_LAYERS + 1];   // waves [first_wave[l], first_wave[l + 1]) work on layer l
    int32_t rows_per_wave[VF_MLP_MAX_LAYERS];    // even
    int64_t part_off[VF_MLP_MAX_LAYERS];          // float offset of the layer's partial block: waves x (K No + No)
};

// VA / VB: the layer's No / K is exactly NT / KT full tiles and the rows are 16-byte aligned: a lane then loads NT
// (KT) CONSECUTIVE columns of its row with one vector load (32 lanes = one contiguous 128 NT bytes) and feeds component
// i to tile i -- the (tile, lane) -> column assignment is a free choice, it only permutes where dW lands in the
// accumulators: n = NT lane + i instead of 32 i + lane.
template <int W>
__device__ __forceinline__ void wgrad_load(__amdgpu_buffer_rsrc_t r, unsigned off, float (&f)[W])
{
    if constexpr (W == 4) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);      // (__builtin_bit_cast of a vector ELEMENT reads element 0)
    } else {
        static_assert(W == 2, "vector operand loads: 2 or 4 tiles");
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
        f[0] = __uint_as_float(v[0]);
        f[1] = __uint_as_float(v[1]);
    }
}

// Operand rows through BUFFER loads whose descriptors cover exactly the wave's slab [r0, r1): a row past the slab reads as zeros
// (hardware range check), so the steps that overhang the slab need neither clamped row indices nor `live` multipliers, and an
// address is a 32-bit lane offset + one add per step instead of a 64-bit multiply-add per operand (r03: ~27 VALU instructions per
// step next to its NT x KT MFMAs, and hipcc piled the address arithmetic of a whole unrolled body ahead of its MFMAs).
template <int NT, int KT, bool VA, bool VB, bool SMALL>
__device__ __forceinline__ void wgrad_slab(const vf_mlp_bwd_layer& L, int r0, int r1, float* __restrict__ part)
{
    constexpr int kWgDepth = wg_depth(NT, KT, SMALL);
    const int lane = threadIdx.x & 63, c = lane & 31, kk = lane >> 5;
    f32x16 acc[NT][KT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[i][j] = f32x16{0};
    float bsum[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) bsum[i] = 0.0f;
    const int rows = r1 - r0;
    float* pa = const_cast<float*>(L.dY) + (size_t)r0 * L.ld_dy;
    float* pb = const_cast<float*>(L.X) + (size_t)r0 * L.ld_x;
    const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc(pa, 0, ((rows - 1) * L.ld_dy + L.No) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc(pb, 0, ((rows - 1) * L.ld_x + L.K) * 4, 0x00020000);
    // scalar mode: column guards hoisted (clamped column + multiplier 0 / 1)
    unsigned an[NT], bk[KT];
    float am[NT], bm[KT];
#pragma unroll
    for (int i = 0; i < NT; ++i) { const int n = 32 * i + c; an[i] = 4u * (unsigned)(n < L.No ? n : L.No - 1); am[i] = n < L.No ? 1.0f : 0.0f; }
#pragma unroll
    for (int j = 0; j < KT; ++j) { const int k = 32 * j + c; bk[j] = 4u * (unsigned)(k < L.K ? k : L.K - 1); bm[j] = k < L.K ? 1.0f : 0.0f; }
    float ra[kWgDepth][NT], rb[kWgDepth][KT];
    const int steps = (rows + 1) >> 1;
    // byte offsets of this lane's row of the step being issued (row 2 s + kk of the slab); + 2 rows per step
    unsigned oa = (unsigned)kk * (unsigned)L.ld_dy * 4u + (VA ? (unsigned)(NT * c) * 4u : 0u);
    unsigned ob = (unsigned)kk * (unsigned)L.ld_x * 4u + (VB ? (unsigned)(KT * c) * 4u : 0u);
    const unsigned da = 8u * (unsigned)L.ld_dy, db = 8u * (unsigned)L.ld_x;
    auto issue = [&](float (&fa)[NT], float (&fb)[KT]) {
        if constexpr (VA) {
            wgrad_load<NT>(ra_src, oa, fa);
        } else {
#pragma unroll
            for (int i = 0; i < NT; ++i) fa[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra_src, (int)(oa + an[i]), 0, 0));
        }
        if constexpr (VB) {
            wgrad_load<KT>(rb_src, ob, fb);
        } else {
#pragma unroll
            for (int j = 0; j < KT; ++j) fb[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb_src, (int)(ob + bk[j]), 0, 0));
        }
        oa += da;
        ob += db;
    };
#pragma unroll
    for (int p = 0; p < kWgDepth; ++p) issue(ra[p], rb[p]);      // rows past the slab: zeros
    for (int s0 = 0; s0 < steps; s0 += kWgDepth) {     // branch-free body: steps past the slab multiply zeros
#pragma unroll
        for (int p = 0; p < kWgDepth; ++p) {
            float fa[NT], fb[KT];
#pragma unroll
            for (int i = 0; i < NT; ++i) { fa[i] = VA ? ra[p][i] : ra[p][i] * am[i]; bsum[i] += fa[i]; }
#pragma unroll
            for (int j = 0; j < KT; ++j) fb[j] = VB ? rb[p][j] : rb[p][j] * bm[j];
            issue(ra[p], rb[p]);
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < KT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            // keep the steps in program order: the scheduler otherwise clusters all loads of the unrolled body at its
            // top and drains them (vmcnt(0)) by its end, which collapses the prefetch distance to less than one body
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // partial in accumulator order (every store instruction writes 256 contiguous bytes): tile (i, j), register r, lane;
    // then the bias sums [NT][32].  k_wgrad_fold maps the positions back to (n, k) with wgrad_nk().
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[((i * KT + j) * 16 + r) * 64 + lane] = acc[i][j][r];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const float other = __shfl_xor(bsum[i], 32);
        if (kk == 0) part[NT * KT * 1024 + 32 * i + c] = bsum[i] + other;
    }
}

__host__ __device__ inline bool wgrad_vec_ok(const float* p, int ld, int w, int tiles)
{
    return (tiles == 2 || tiles == 4) && w == 32 * tiles && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}

// floats of one wave's partial of this layer
__host__ __device__ inline int wgrad_partial_size(const vf_mlp_bwd_layer& L)
{
    const int NT = (L.No + 31) >> 5, KT = (L.K + 31) >> 5;
    return NT * KT * 1024 + NT * 32;
}

// element e of a partial -> parameter offset in [dW (No x K) | db (No)], or -1 (padding)
__device__ __forceinline__ int wgrad_param_of(const vf_mlp_bwd_layer& L, int e)
{
    const int NT = (L.No + 31) >> 5, KT = (L.K + 31) >> 5;
    const bool va = wgrad_vec_ok(L.dY, L.ld_dy, L.No, NT), vb = wgrad_vec_ok(L.X, L.ld_x, L.K, KT);
    if (e >= NT * KT * 1024) {
        const int b = e - NT * KT * 1024, i = b >> 5, c = b & 31, n = va ? NT * c + i : 32 * i + c;
        return n < L.No ? L.K * L.No + n : -1;
    }
    const int lane = e & 63, r = (e >> 6) & 15, t = e >> 10, i = t / KT, j = t - i * KT, c = lane & 31, kk = lane >> 5;
    const int ia = 4 * kk + (r & 3) + 8 * (r >> 2);
    const int n = va ? NT * ia + i : 32 * i + ia, k = vb ? KT * c + j : 32 * j + c;
    return (n < L.No && k < L.K) ? n * L.K + k : -1;
}

template <int NT, int KT, bool SMALL>
__device__ __forceinline__ void wgrad_slab_pick(const vf_mlp_bwd_layer& L, int r0, int r1, float* __restrict__ part)
{
    const bool va = wgrad_vec_ok(L.dY, L.ld_dy, L.No, NT), vb = wgrad_vec_ok(L.X, L.ld_x, L.K, KT);
    if constexpr ((NT == 2 || NT == 4) && (KT == 2 || KT == 4)) {
        if (va && vb) return wgrad_slab<NT, KT, true, true, SMALL>(L, r0, r1, part);
    }
    if constexpr (NT == 2 || NT == 4) {
        if (va) return wgrad_slab<NT, KT, true, false, SMALL>(L, r0, r1, part);
    }
    if constexpr (KT == 2 || KT == 4) {
        if (vb) return wgrad_slab<NT, KT, false, true, SMALL>(L, r0, r1, part);
    }
    wgrad_slab<NT, KT, false, false, SMALL>(L, r0, r1, part);
}

// SMALL: every layer of the table has at most 8 accumulator tiles (the reference-default policies: 128 -> 64 is the largest layer), so
// a wave fits 256 VGPRs and TWO waves share a SIMD -- the launch streams X / dZ and is bound by how much of that is in flight
template <bool SMALL>
__global__ __launch_bounds__(64, SMALL ? 2 : 1) void k_mlp_wgrad(const vf_mlp_bwd_desc d, const WgradTable t, float* __restrict__ partials, int M)
{
    prefetch_kernarg<sizeof(vf_mlp_bwd_desc) + sizeof(WgradTable) + 16>();
    const int w = blockIdx.x;
    int l = 0;
    while (l + 1 < t.n_layers && w >= t.first_wave[l + 1]) ++l;
    const vf_mlp_bwd_layer& L = d.layer[l];
    const int lw = w - t.first_wave[l];
    const int r0 = lw * t.rows_per_wave[l], r1 = min(r0 + t.rows_per_wave[l], M);
    float* part = partials + t.part_off[l] + (size_t)lw * wgrad_partial_size(L);
    const int NT = (L.No + 31) >> 5, KT = (L.K + 31) >> 5;
    if (r0 >= r1) {        // empty slab (rounding): the fold still reads this partial
        for (int i = threadIdx.x; i < wgrad_partial_size(L); i += 64) part[i] = 0.0f;
        return;
    }
    switch (NT * 4 + KT - 5) {
    case 0: wgrad_slab_pick<1, 1, SMALL>(L, r0, r1, part); break;
    case 1: wgrad_slab_pick<1, 2, SMALL>(L, r0, r1, part); break;
    case 2: wgrad_slab_pick<1, 3, SMALL>(L, r0, r1, part); break;
    case 3: wgrad_slab_pick<1, 4, SMALL>(L, r0, r1, part); break;
    case 4: wgrad_slab_pick<2, 1, SMALL>(L, r0, r1, part); break;
    case 5: wgrad_slab_pick<2, 2, SMALL>(L, r0, r1, part); break;
    case 6: wgrad_slab_pick<2, 3, SMALL>(L, r0, r1, part); break;
    case 7: wgrad_slab_pick<2, 4, SMALL>(L, r0, r1, part); break;
    case 8: wgrad_slab_pick<3, 1, SMALL>(L, r0, r1, part); break;
    case 9: wgrad_slab_pick<3, 2, SMALL>(L, r0, r1, part); break;
    case 12: wgrad_slab_pick<4, 1, SMALL>(L, r0, r1, part); break;
    case 13: wgrad_slab_pick<4, 2, SMALL>(L, r0, r1, part); break;
    default:
        if constexpr (!SMALL) {
            switch (NT * 4 + KT - 5) {
            case 10: wgrad_slab_pick<3, 3, SMALL>(L, r0, r1, part); break;
            case 11: wgrad_slab_pick<3, 4, SMALL>(L, r0, r1, part); break;
            case 14: wgrad_slab_pick<4, 3, SMALL>(L, r0, r1, part); break;
            default: wgrad_slab_pick<4, 4, SMALL>(L, r0, r1, part); break;
            }
        }
        break;
    }
}

// grad (+)= sum over the layer's waves of partial[wave][e]; 64 consecutive partial elements per block, the 4 waves of
// the block split the partial rows (8 loads in flight each) and combine through LDS in a fixed order
__global__ __launch_bounds__(kBlock) void k_wgrad_fold(const vf_mlp_bwd_desc d, const WgradTable t, const float* __restrict__ partials,
                                                       float* __restrict__ grad, int accumulate, double* __restrict__ sq_part,
                                                       const vf_stats_fold ls, int n_param_blocks)
{
    prefetch_kernarg<sizeof(vf_mlp_bwd_desc) + sizeof(WgradTable) + 40 + sizeof(vf_stats_fold)>();
    __shared__ float red[4][64];
    if ((int)blockIdx.x >= n_param_blocks) {     // the extra block: loss-statistic partial rows (vf_ppo_update) -> stats
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (int k = w; k < 16; k += 4) {         // 64 lanes stride over the rows, shuffle tree: the order of k_fold_stats
            float s = 0.0f;
            if (k < 9) {
                // rows are at most 1024 (vf_ppo_update's contract): all of a lane's loads are issued before the first add
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int b = lane + 64 * i;
                    v[i] = b < ls.n_rows ? ls.part[(size_t)b * 16 + k] : 0.0f;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) s += (lane + 64 * i < ls.n_rows) ? v[i] : 0.0f;
                for (int b = lane + 1024; b < ls.n_rows; b += 64) s += ls.part[(size_t)b * 16 + k];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
            if (lane == 0) {
                ls.stats[k] = s;
                if (ls.d_log_std_out && k >= 5 && k < 9) ls.d_log_std_out[k - 5] = s;
                if (ls.stats_accum) ls.stats_accum[k] += s;
            }
        }
        return;
    }
    int b = blockIdx.x, l = 0;
    for (; l < t.n_layers; ++l) {
        const int nb = (wgrad_partial_size(d.layer[l]) + 63) / 64;
        if (b < nb) break;
        b -= nb;
    }
    if (l >= t.n_layers) return;
    const vf_mlp_bwd_layer& L = d.layer[l];
    const int tot = wgrad_partial_size(L), waves = t.first_wave[l + 1] - t.first_wave[l];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6, e = b * 64 + lane;
    const int prm = e < tot ? wgrad_param_of(L, e) : -1;
    float s = 0.0f;
    if (prm >= 0) {
        const float* p = partials + t.part_off[l] + e;
        float s4[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int w = q;
        for (; w + 28 < waves; w += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s4[u] += p[(size_t)(w + 4 * u) * tot];
        }
        for (; w < waves; w += 4) s4[0] += p[(size_t)w * tot];
        s = ((s4[0] + s4[1]) + (s4[2] + s4[3])) + ((s4[4] + s4[5]) + (s4[6] + s4[7]));
    }
    red[q][lane] = s;
    __syncthreads();
    double sq = 0.0;
    if (q == 0 && prm >= 0) {
        const float v = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
        const int nw = L.K * L.No;
        float* g = grad + (prm < nw ? L.w_off + prm : L.b_off + (prm - nw));
        const float nv = accumulate ? *g + v : v;
        *g = nv;
        sq = (double)nv * (double)nv;
    }
    if (sq_part && q == 0) {       // squared norm of what this block wrote: fixed-order wave sum
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
        if (lane == 0) sq_part[blockIdx.x] = sq;
    }
}

// two waves per SIMD (k_mlp_wgrad<true>: no layer needs more than 8 accumulator tiles)?  At row counts whose operands fit the 256 MiB
// Infinity Cache (the PPO minibatch: 25 600 rows x 5.7 KB) the launch is bound by MFMA issue + load latency and a second wave per SIMD
// only doubles the partials (slower: profiles/r04_ppo_wgrad.txt).  Past it -- SHAC's critic over the 524 288 rows of the horizon buffer,
// 2 GB of X / dZ per launch -- a lone wave per SIMD keeps too few bytes in flight (2.9 TB/s); two waves: 0.70 -> 0.41 ms
// (profiles/r04_shac.txt).  VISFLY_AMD_WGRAD_WPS=1/2 forces the choice (A/B).
constexpr int kWgradStreamRows = 131072;
bool wgrad_small(const vf_mlp_bwd_desc& d, int M)
{
    static const int forced = [] { const char* e = getenv("VISFLY_AMD_WGRAD_WPS"); return e ? atoi(e) : 0; }();
    if (forced == 1 || (forced != 2 && M < kWgradStreamRows)) return false;
    for (int l = 0; l < d.n_layers; ++l)
        if (((d.layer[l].No + 31) >> 5) * ((d.layer[l].K + 31) >> 5) > 8) return false;
    return true;
}

// waves per layer proportional to its MFMA count per row pair; -> total waves, partial floats
int64_t wgrad_plan(const vf_mlp_bwd_desc& d, int M, WgradTable& t, int* total_waves)
{
    int tiles[VF_MLP_MAX_LAYERS], sum = 0;
    for (int l = 0; l < d.n_layers; ++l) {
        tiles[l] = ((d.layer[l].No + 31) >> 5) * ((d.layer[l].K + 31) >> 5) + 2;   // + per-row-pair overhead (loads, guards) in MFMA units
        sum += tiles[l];
    }
    const int budget = wgrad_small(d, M) ? 2048 : 1024;   // waves per SIMD x 1024 SIMDs
    t.n_layers = d.n_layers;
    int w = 0;
    int64_t off = 0;
    for (int l = 0; l < d.n_layers; ++l) {
        int nw = (int)(((int64_t)budget * tiles[l] + sum / 2) / sum);
        if (nw < 1) nw = 1;
        int rows = (M + nw - 1) / nw;
        rows = (rows + 1) & ~1;
        if (rows < 2) rows = 2;
        nw = (M + rows - 1) / rows;
        t.first_wave[l] = w;
        t.rows_per_wave[l] = rows;
        t.part_off[l] = off;
        w += nw;
        off += (int64_t)nw * wgrad_partial_size(d.layer[l]);
    }
    t.first_wave[d.n_layers] = w;
    *total_waves = w;
    return off;
}

int64_t mlp_wgrad_partial_floats(const vf_mlp_bwd_desc* d, int M)
{
    WgradTable t;
    int w;
    return wgrad_plan(*d, M, t, &w);
}

int mlp_wgrad_fold_blocks(const vf_mlp_bwd_desc* d)
{
    int nb = 0;
    for (int l = 0; l < d->n_layers; ++l) nb += (wgrad_partial_size(d->layer[l]) + 63) / 64;
    return nb;
}

int mlp_wgrad_launch(const vf_mlp_bwd_desc* d, float* partials, float* grad, int M, int accumulate, double* sq_part,
                     const vf_stats_fold* loss_stats, hipStream_t st)
{
    WgradTable t;
    int waves = 0;
    wgrad_plan(*d, M, t, &waves);
    if (wgrad_small(*d, M)) hipLaunchKernelGGL(k_mlp_wgrad<true>, dim3(waves), dim3(64), 0, st, *d, t, partials, M);
    else hipLaunchKernelGGL(k_mlp_wgrad<false>, dim3(waves), dim3(64), 0, st, *d, t, partials, M);
    const int nb = mlp_wgrad_fold_blocks(d);
    const vf_stats_fold ls = loss_stats ? *loss_stats : vf_stats_fold{};
    hipLaunchKernelGGL(k_wgrad_fold, dim3(nb + (loss_stats ? 1 : 0)), dim3(kBlock), 0, st, *d, t, (const float*)partials, grad, accumulate,
                       sq_part, ls, nb);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

}  // namespace vf
