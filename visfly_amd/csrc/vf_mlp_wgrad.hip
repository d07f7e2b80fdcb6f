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
#include "vf_adam_device.hpp"

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

// What one wave writes and another reads within the launch (the fused tail's partials, gradients, per-block sums) goes out as agent-scope
// stores -- write-through to where the eight XCDs' L2s agree -- and comes back through agent-scope loads: no L2 write-back / invalidate
// fences between the waves.  The partials are written that way in every form of the launch (the separate fold reads them from another
// launch: nothing changes for it, and the launch ends without 20 MB of dirty lines behind it).
template <typename T>
__device__ __forceinline__ void store_agent(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ T load_agent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

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
            for (int r = 0; r < 16; ++r) store_agent(part + ((i * KT + j) * 16 + r) * 64 + lane, acc[i][j][r]);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const float other = __shfl_xor(bsum[i], 32);
        if (kk == 0) store_agent(part + NT * KT * 1024 + 32 * i + c, bsum[i] + other);
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

// ---- fold order (shared by k_wgrad_fold and the fused tail) ------------------------------------------------------------------------
// element e of the layer's gradient = sum over the layer's waves of partial[w][e], in 32 chains: chain (q, u), q = 0..3, u = 0..7, takes
// the partial rows w = q + 4 u + 32 k in ascending k; a q's chains combine as ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)), the four
// q's as ((r0 + r1) + r2) + r3.  In k_wgrad_fold q is the wave of the block (the four combine through LDS); in the fused tail one lane
// walks all 32 chains with rows past the layer read as +0 (adding +0 to a chain that started at +0 changes no bit).
#ifndef VF_FOLD_BATCH
#define VF_FOLD_BATCH 1      // rounds of 8 partial rows fetched before the first add (A/B knob: 2 / 4 / 8 rounds = 16 / 32 / 64 loads in flight per lane
                             // measured 118.8 / 119.6 / 122.3 us per optimiser step against 118.7 for 1 on the same box -- profiles/r06_fused_tail.txt, 7)
#endif
__device__ __forceinline__ float fold_chains_q(const float* __restrict__ p, size_t tot, int waves, int q)
{
    float s4[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int w = q;
    // VF_FOLD_BATCH rounds are fetched before their adds -- the same adds in the same order (chain u takes rows w + 4 u + 32 k in
    // ascending k): same bits.  More loads in flight per lane do NOT shorten the fold (r06: 800 blocks x 4 waves already cover the latency)
    // (rows past the layer's block are read as +0: a chain that started at +0 is never -0, so adding +0 changes no bit)
    for (; w < waves; w += 32 * VF_FOLD_BATCH) {
        float v[VF_FOLD_BATCH][8];
#pragma unroll
        for (int k = 0; k < VF_FOLD_BATCH; ++k)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = w + 32 * k + 4 * u;
                v[k][u] = row < waves ? p[(size_t)row * tot] : 0.0f;
            }
#pragma unroll
        for (int k = 0; k < VF_FOLD_BATCH; ++k)
#pragma unroll
            for (int u = 0; u < 8; ++u) s4[u] += v[k][u];
    }
    return ((s4[0] + s4[1]) + (s4[2] + s4[3])) + ((s4[4] + s4[5]) + (s4[6] + s4[7]));
}

// statistic k of the loss-statistic partial rows (vf_ppo_update's scratch): 64 lanes stride over the rows, shuffle tree; lane 0 holds the
// sum -- the order of k_fold_stats.  ROWS16: the caller's rows already in registers (fused tail: all statistics loaded in one batch)
__device__ __forceinline__ float stats_fold_tree(float s)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    return s;
}
__device__ __forceinline__ float stats_fold_one(const vf_stats_fold& ls, int k, int lane)
{
    float s = 0.0f;
    if (k < 9) {
        // rows are at most 1024 in the common case (vf_ppo_update at <= 32 768 rows): all of a lane's loads are issued before the first add
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
    return stats_fold_tree(s);
}

// ---- fused tail: fold -> squared norm -> clip -> Adam -> packed-weight refresh inside the weight-gradient launch ----------------------
// The four launches of an optimiser step (chain, weight gradients, fold, Adam) were 105 us of kernels in 112 us: the fold (9.4 us) and Adam
// (5.3 us) are latency, and every launch boundary costs ~1.8 us behind the dirty lines of the one before.  A lone workgroup cannot fold (the
// partials are 20 MB: one wave reads ~6 GB/s), so the fold stays chip-wide: the waves of a LAYER meet once their partials are out, every
// wave then folds 64 elements of its own layer (all of its up-to-160 loads in flight at once), writes the gradient and the block's fp64 sum
// of squares; all waves meet once more, add up the per-block sums in k_adam's order and run Adam on the elements they still hold.  Same
// reduction orders as the separate launches: the two paths agree to the bit.
// What a meeting of ~1 000 lone waves costs on gfx950 (tools/grid_barrier_probe.hip, profiles/r06_fused_tail.txt): every wave adding to ONE
// word and polling ONE word 62 us (same-address agent-scope atomics and polls are served one by one, ~10 ns each); an agent-scope release
// fence (buffer_wbl2) 7 us when all waves issue it; arrivals grouped eight ways (workgroup id mod 8), the group's last arrival carrying it to
// the global word, release handed back the same way, and NO fences -- everything that crosses waves is written with agent-scope
// (write-through) stores, waited for (vmcnt) and read with agent-scope loads -- 3.5 us.  That is the form used here.
// The launch needs every wave co-resident (1 024 waves of 512 VGPRs = one per SIMD): the host checks the plan against the occupancy of this
// kernel on this device and answers VF_EUNSUPPORTED otherwise; a wave that waits longer than `timeout` ticks of the 100 MHz clock raises
// the abort word and every waiter leaves (the caller then sees VF_WGRAD_SYNC_ABORT set: hard error, no hang).
struct WgradTail {
    float* grad;
    float* param;
    float* m;
    float* v;
    double* sq_part;          // [n_fold_blocks]
    unsigned* sync;           // slots of 16 words (one 64-byte line each), see tail_slot
    vf_adam_cfg adam;
    float step, bc2_sqrt;
    vf_stats_fold ls;
    int n, accumulate, n_fold_blocks, has_ls;
    long long timeout;
    long long* trace;         // VF_WGRAD_TRACE builds only
};

// slot 0: generation (= release word of the grid meeting), 1: abort (VF_WGRAD_SYNC_ABORT = 16), 2: grid meeting's global count,
// 3..10 its group counts, 11..18 its group flags; layer l: 19 + 18 l: global count, + 1 release word, + 2..9 group counts, + 10..17 group flags
__device__ __forceinline__ unsigned* tail_slot(unsigned* sync, int i) { return sync + 16 * i; }
static_assert((19 + 18 * VF_MLP_MAX_LAYERS) * 16 <= VF_WGRAD_SYNC_WORDS, "sync words");
static_assert(VF_WGRAD_SYNC_ABORT == 16, "abort word = slot 1");

__device__ __forceinline__ unsigned tail_peek(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tail_poke(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// this wave's agent-scope stores have been acknowledged (and the compiler keeps what follows behind them); no L2 write-back: they were
// write-through.  The mirror image after a wait: what follows is not hoisted above the poll
__device__ __forceinline__ void tail_stores_out() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
__device__ __forceinline__ void tail_after_wait() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

// release words and group flags are 64-bit: the generation in the low half, a payload in the high half (the grid meeting hands the clip
// coefficient to everybody with the flag itself: no wave but one reads the per-block sums)
__device__ __forceinline__ unsigned long long tail_peek64(const unsigned* p)
{
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tail_poke64(unsigned* p, unsigned gen, unsigned payload)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)payload << 32) | gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// -> false: aborted (time limit of this or of another wave); else *payload = the high half of the word once its low half is `gen`
__device__ __forceinline__ bool tail_wait_for(const unsigned* word, unsigned gen, unsigned* payload, unsigned* abort_word, long long timeout)
{
    const long long t0 = wall_clock64();
    unsigned long long v = tail_peek64(word);
    for (int it = 1; (unsigned)v != gen; ++it) {
        __builtin_amdgcn_s_sleep(1);
        if ((it & 31) == 0) {
            if (tail_peek(abort_word)) return false;
            if (wall_clock64() - t0 > timeout) {
                tail_poke(abort_word, 1u);
                return false;
            }
        }
        v = tail_peek64(word);
    }
    *payload = (unsigned)(v >> 32);
    return true;
}

// members of [f0, f1) whose workgroup id is x mod 8
__device__ __forceinline__ unsigned tail_group_size(int f0, int f1, int x) { return (unsigned)(((f1 - x + 7) >> 3) - ((f0 - x + 7) >> 3)); }

// One meeting of the workgroups [f0, f1): arrive at the group's count (group = workgroup id mod 8); the group's last arrival carries it to
// the global count; the last group there runs `final_fn` (-> payload) and raises the release word to (gen, payload); the groups' last
// arrivals wait for that and raise their group's flag, everybody else waits for the flag.  Counts return to zero through `final_fn` of the
// launch's last meeting.  -> false: aborted
template <typename F>
__device__ __forceinline__ bool tail_meet(unsigned* global, unsigned* release, unsigned* gcount, unsigned* gflag, int f0, int f1, unsigned gen,
                                          unsigned* payload, unsigned* abort_word, long long timeout, F final_fn)
{
    const int x = blockIdx.x & 7;
    unsigned groups = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) groups += tail_group_size(f0, f1, g) > 0 ? 1u : 0u;
    unsigned old = 0;
    if ((threadIdx.x & 63) == 0) old = __hip_atomic_fetch_add(gcount + 16 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old + 1u == tail_group_size(f0, f1, x)) {
        unsigned old2 = 0;
        if ((threadIdx.x & 63) == 0) old2 = __hip_atomic_fetch_add(global, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old2 = __builtin_amdgcn_readfirstlane(old2);
        if (old2 + 1u == groups) {
            *payload = final_fn();
            tail_stores_out();
            tail_poke64(release, gen, *payload);
        } else if (!tail_wait_for(release, gen, payload, abort_word, timeout)) {
            return false;
        }
        tail_poke64(gflag + 16 * x, gen, *payload);
    } else if (!tail_wait_for(gflag + 16 * x, gen, payload, abort_word, timeout)) {
        return false;
    }
    tail_after_wait();
    return true;
}

// -DVF_WGRAD_TRACE (tools/build_variant.py; never in the shipped library): every wave of the fused launch leaves the 100 MHz clock at
// its phase boundaries in the 8 words behind its partial's sq_part... -- a buffer the experiment passes through VISFLY_AMD_WGRAD_TRACE_PTR
#ifdef VF_WGRAD_TRACE
#define VF_TRACE_MARK(k) do { if (trace && (threadIdx.x & 63) == 0) trace[8 * blockIdx.x + (k)] = wall_clock64(); } while (0)
#else
#define VF_TRACE_MARK(k) do { } while (0)
#endif
constexpr int kAuxSc1 = 16;       // cache-policy operand of the buffer builtins on gfx940+: bit 4 = sc1 (agent scope: coherent across the XCDs' L2s)
constexpr int kTailBatch = 5;      // partial rows fetched per round of the in-kernel fold: 32 x 5 = 160 loads in flight per lane

__device__ __forceinline__ float tail_fold_element(__amdgpu_buffer_rsrc_t rs, unsigned e, unsigned tot, int waves)
{
    float s[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < 8; ++u) s[q][u] = 0.0f;
    for (int k0 = 0; 32 * k0 < waves; k0 += kTailBatch) {
        float v[kTailBatch][4][8];
#pragma unroll
        for (int k = 0; k < kTailBatch; ++k)
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) {       // rows past the layer's block: the descriptor's range check answers +0
                    const unsigned w = 32u * (unsigned)(k0 + k) + 4u * u + q;
                    v[k][q][u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)((w * tot + e) * 4u), 0, kAuxSc1));
                }
#pragma unroll
        for (int k = 0; k < kTailBatch; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int u = 0; u < 8; ++u) s[q][u] += v[k][q][u];
    }
    float r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = ((s[q][0] + s[q][1]) + (s[q][2] + s[q][3])) + ((s[q][4] + s[q][5]) + (s[q][6] + s[q][7]));
    return ((r[0] + r[1]) + r[2]) + r[3];
}

__device__ __forceinline__ double bcast_lane0(double x)
{
    const unsigned long long b = __double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ void wgrad_tail(const vf_mlp_bwd_desc& d, const WgradTable& t, float* __restrict__ partials, const WgradTail& T,
                                        int l, int lw)
{
    const int lane = threadIdx.x & 63;
    const int W = t.first_wave[l + 1] - t.first_wave[l], total = t.first_wave[t.n_layers];
    const bool tail_wave = (int)blockIdx.x == total - 1;         // folds the loss statistics, owns the parameters the layers do not cover
    unsigned* const abort_word = tail_slot(T.sync, 1);
#ifdef VF_WGRAD_TRACE
    long long* const trace = T.trace;
#endif
    VF_TRACE_MARK(1);                     // slab done, partial stores issued
    const unsigned gen = tail_peek(tail_slot(T.sync, 0)) + 1u;      // (slot 0 = the grid meeting's release word: generation | clip coefficient)
    unsigned* const lay = tail_slot(T.sync, 19 + 18 * l);
    // (1) this wave's partial is out (agent-scope stores, acknowledged): meet the layer's other waves -- arrival now, the wait after the
    // loss statistics
    tail_stores_out();
    unsigned none;
    if (!tail_meet(lay, lay + 16, lay + 32, lay + 160, t.first_wave[l], t.first_wave[l + 1], gen, &none, abort_word, T.timeout, [] { return 0u; })) return;
    VF_TRACE_MARK(2);                     // the layer's partials are all out
    // (2) fold: block b of the layer = 64 consecutive elements of its partial row, as k_wgrad_fold's block
    const vf_mlp_bwd_layer& L = d.layer[l];
    const int tot = wgrad_partial_size(L), nb_l = (tot + 63) / 64;
    int blk0 = 0;
    for (int i = 0; i < l; ++i) blk0 += (wgrad_partial_size(d.layer[i]) + 63) / 64;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(partials + t.part_off[l], 0, (int)((size_t)W * tot * 4), 0x00020000);
    const int nw = L.K * L.No;
    float g0 = 0.0f, p0 = 0.0f, m0 = 0.0f, v0 = 0.0f;
    long i0 = -1;
    for (int b = lw; b < nb_l; b += W) {
        const int e = b * 64 + lane;
        const int prm = e < tot ? wgrad_param_of(L, e) : -1;
        const long idx = prm < 0 ? -1 : prm < nw ? L.w_off + prm : L.b_off + (prm - nw);
        float gold = 0.0f;
        if (idx >= 0 && T.accumulate) gold = T.grad[idx];
        if (b == lw && idx >= 0) { i0 = idx; p0 = T.param[idx]; m0 = T.m[idx]; v0 = T.v[idx]; }     // requested ahead of the fold's loads
        const float vsum = tail_fold_element(rs, (unsigned)(e < tot ? e : 0), (unsigned)tot, W);
        double sq = 0.0;
        if (idx >= 0) {
            const float nv = T.accumulate ? gold + vsum : vsum;
            T.grad[idx] = nv;
            sq = (double)nv * (double)nv;
            if (b == lw) g0 = nv;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
        if (lane == 0) store_agent(T.sq_part + blk0 + b, sq);
    }
    if (tail_wave && T.has_ls) {          // while the others fold (the last wave of the last layer seldom has a block of its own): the loss
        // statistics (rows written by the launch before this one); d_log_std goes into the gradient's uncovered tail
        float st[16];
        if (T.ls.n_rows <= 1024) {
            float4 v[16][3];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int b = lane + 64 * i;
                const float4* row = reinterpret_cast<const float4*>(T.ls.part + (size_t)(b < T.ls.n_rows ? b : 0) * 16);
#pragma unroll
                for (int c = 0; c < 3; ++c) v[i][c] = row[c];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                float s = 0.0f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float4 x = v[i][k >> 2];
                    const float xv = (k & 3) == 0 ? x.x : (k & 3) == 1 ? x.y : (k & 3) == 2 ? x.z : x.w;
                    s += (lane + 64 * i < T.ls.n_rows) ? xv : 0.0f;
                }
                st[k] = stats_fold_tree(s);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) st[k] = stats_fold_one(T.ls, k, lane);
        }
#pragma unroll
        for (int k = 9; k < 16; ++k) st[k] = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float s = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(st[k])));
            if (lane == 0) {
                T.ls.stats[k] = s;
                if (T.ls.d_log_std_out && k >= 5 && k < 9) store_agent(T.ls.d_log_std_out + (k - 5), s);
                if (T.ls.stats_accum) T.ls.stats_accum[k] += s;
            }
        }
    }
    // the uncovered parameters [sumsq_tail_from, n): their gradients are the loss statistics' d_log_std (folded above) or what the caller
    // left in grad; the tail wave owns them
    const int tail_from = T.adam.sumsq_tail_from;
    VF_TRACE_MARK(3);                     // own block folded, gradient and block sum issued
    // (3) all gradients and per-block sums are out: meet everybody.  The last arrival of all returns every count to zero (each wave is
    // past its layer's meeting by then), forms the squared norm and raises the generation -- this meeting's release word -- with the
    // clip coefficient in its upper half
    tail_stores_out();
    unsigned coef_bits;
    if (!tail_meet(tail_slot(T.sync, 2), tail_slot(T.sync, 0), tail_slot(T.sync, 3), tail_slot(T.sync, 11), 0, total, gen, &coef_bits, abort_word, T.timeout, [&] {
            for (int i = lane; i < 9; i += 64) tail_poke(tail_slot(T.sync, 2 + i), 0u);
            for (int i = lane; i < 18 * t.n_layers; i += 64)
                if (i % 18 != 1 && i % 18 < 10) tail_poke(tail_slot(T.sync, 19 + i), 0u);
            // (4) squared norm, by this one wave for everybody: k_adam's order -- "thread" 64 q + lane of its 256 sums the partials
            // 64 q + lane + 256 j, then the uncovered tail; four wave sums, ((0 + 1) + (2 + 3))
            float coef = 1.0f;
            if (T.adam.max_grad_norm > 0.0f) {
                tail_after_wait();
                double a[4], x[4][4], xt[4];
                // every load of the common case (<= 1 024 blocks, <= 256 uncovered parameters) is requested before the first sum
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = 64 * q + lane + 256 * j;
                        x[q][j] = i < T.n_fold_blocks ? load_agent(T.sq_part + i) : 0.0;
                    }
                    const long it = tail_from + 64 * q + lane;
                    xt[q] = it < T.n ? (double)load_agent(T.grad + it) : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a[q] = ((0.0 + x[q][0]) + x[q][1]) + x[q][2];
                    a[q] += x[q][3];
                    for (int i = 64 * q + lane + 1024; i < T.n_fold_blocks; i += 256) a[q] += load_agent(T.sq_part + i);
                    a[q] += xt[q] * xt[q];
                    for (long i = tail_from + 64 * q + lane + 256; i < T.n; i += 256) {
                        const double gi = (double)load_agent(T.grad + i);
                        a[q] += gi * gi;
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) a[q] += __shfl_down(a[q], o, 64);
                    a[q] = bcast_lane0(a[q]);
                }
                coef = adam_clip_coef((float)((a[0] + a[1]) + (a[2] + a[3])), T.adam.max_grad_norm);
            }
            return __float_as_uint(coef);
        }))
        return;
    VF_TRACE_MARK(4);                     // everybody's block sums are out, the clip coefficient came with the flag
    const float coef = __uint_as_float(coef_bits);
    VF_TRACE_MARK(5);                     // clip coefficient known
    // (5) Adam on the elements this wave folded (+ the uncovered tail on the tail wave)
    if (i0 >= 0) {
        const float pn = adam_param(p0, g0, m0, v0, coef, T.adam, T.step, T.bc2_sqrt);
        T.m[i0] = m0;
        T.v[i0] = v0;
        T.param[i0] = pn;
        if (T.adam.pack_map) adam_refresh_packed(T.adam, i0, pn);
    }
    for (int b = lw + W; b < nb_l; b += W) {
        const int e = b * 64 + lane;
        const int prm = e < tot ? wgrad_param_of(L, e) : -1;
        if (prm < 0) continue;
        const long idx = prm < nw ? L.w_off + prm : L.b_off + (prm - nw);
        float mi = T.m[idx], vi = T.v[idx];
        const float pn = adam_param(T.param[idx], T.grad[idx], mi, vi, coef, T.adam, T.step, T.bc2_sqrt);
        T.m[idx] = mi;
        T.v[idx] = vi;
        T.param[idx] = pn;
        if (T.adam.pack_map) adam_refresh_packed(T.adam, idx, pn);
    }
    if (tail_wave) {
        for (long i = tail_from + lane; i < T.n; i += 64) {
            float mi = T.m[i], vi = T.v[i];
            const float pn = adam_param(T.param[i], load_agent(T.grad + i), mi, vi, coef, T.adam, T.step, T.bc2_sqrt);
            T.m[i] = mi;
            T.v[i] = vi;
            T.param[i] = pn;
            if (T.adam.pack_map) adam_refresh_packed(T.adam, i, pn);
        }
    }
    VF_TRACE_MARK(6);
}

// SMALL: every layer of the table has at most 8 accumulator tiles (the reference-default policies: 128 -> 64 is the largest layer), so
// a wave fits 256 VGPRs and TWO waves share a SIMD -- the launch streams X / dZ and is bound by how much of that is in flight
// TAIL: the fold, the gradient norm, the clip and Adam happen in this launch (wgrad_tail)
template <bool SMALL, bool TAIL>
__global__ __launch_bounds__(64, SMALL ? 2 : 1) void k_mlp_wgrad(const vf_mlp_bwd_desc d, const WgradTable t, float* __restrict__ partials, int M,
                                                                 const WgradTail tail)
{
    prefetch_kernarg<sizeof(vf_mlp_bwd_desc) + sizeof(WgradTable) + 16 + (TAIL ? sizeof(WgradTail) : 0)>();
#ifdef VF_WGRAD_TRACE
    long long* const trace = tail.trace;
#endif
    VF_TRACE_MARK(0);
    const int w = blockIdx.x;
    int l = 0;
    while (l + 1 < t.n_layers && w >= t.first_wave[l + 1]) ++l;
    const vf_mlp_bwd_layer& L = d.layer[l];
    const int lw = w - t.first_wave[l];
    const int r0 = lw * t.rows_per_wave[l], r1 = min(r0 + t.rows_per_wave[l], M);
    float* part = partials + t.part_off[l] + (size_t)lw * wgrad_partial_size(L);
    const int NT = (L.No + 31) >> 5, KT = (L.K + 31) >> 5;
    if (r0 >= r1) {        // empty slab (rounding): the fold still reads this partial
        for (int i = threadIdx.x; i < wgrad_partial_size(L); i += 64) store_agent(part + i, 0.0f);
    } else {
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
    if constexpr (TAIL) wgrad_tail(d, t, partials, tail, l, lw);
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
        for (int k = w; k < 16; k += 4) {
            const float s = stats_fold_one(ls, k, lane);
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
    if (prm >= 0) s = fold_chains_q(partials + t.part_off[l] + e, (size_t)tot, waves, q);
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

// waves per layer proportional to its MFMA count per row pair; -> total waves, partial floats.  The total never exceeds the budget
// (1 024 SIMDs x waves per SIMD): the fused tail needs every wave of the launch resident at once
int64_t wgrad_plan(const vf_mlp_bwd_desc& d, int M, WgradTable& t, int* total_waves)
{
    const bool small = wgrad_small(d, M);
    int tiles[VF_MLP_MAX_LAYERS], sum = 0;
    for (int l = 0; l < d.n_layers; ++l) {
        // cost of a row pair in MFMA units: tiles + the per-row-pair overhead (loads, address steps).  Measured on the wave timeline
        // of the 25 600-row launch (profiles/r06_fused_tail.txt): 116 / 180 / 346 ns per row pair at 2 / 4 / 8 tiles = 38.5 (tiles + 0.9);
        // until r06 the plan said tiles + 2 and the 8-tile layers' waves ran 31.5 us next to 26 us for the 2-tile layers'.  The
        // streaming regime (two waves per SIMD, bandwidth) keeps its + 2
        tiles[l] = ((d.layer[l].No + 31) >> 5) * ((d.layer[l].K + 31) >> 5) + (small ? 2 : 1);
        sum += tiles[l];
    }
    const int cap = small ? 2048 : 1024;   // waves per SIMD x 1024 SIMDs
    t.n_layers = d.n_layers;
    for (int budget = cap;; budget -= 8) {
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
        if (w <= cap || budget <= 8) return off;
    }
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
    if (wgrad_small(*d, M)) hipLaunchKernelGGL((k_mlp_wgrad<true, false>), dim3(waves), dim3(64), 0, st, *d, t, partials, M, WgradTail{});
    else hipLaunchKernelGGL((k_mlp_wgrad<false, false>), dim3(waves), dim3(64), 0, st, *d, t, partials, M, WgradTail{});
    const int nb = mlp_wgrad_fold_blocks(d);
    const vf_stats_fold ls = loss_stats ? *loss_stats : vf_stats_fold{};
    hipLaunchKernelGGL(k_wgrad_fold, dim3(nb + (loss_stats ? 1 : 0)), dim3(kBlock), 0, st, *d, t, (const float*)partials, grad, accumulate,
                       sq_part, ls, nb);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

// the launches of mlp_wgrad_launch for the layers whose bit is set in `layer_mask` only, on the row-slab plan of the WHOLE table: the same
// slabs, partials and fold order per layer as the one launch over all layers -- a gradient formed in two such calls (the two buckets of
// the two-bucket exchange) has the bits of the one formed in one
int mlp_wgrad_launch_layers(const vf_mlp_bwd_desc* d, float* partials, float* grad, int M, int accumulate, unsigned layer_mask, hipStream_t st)
{
    WgradTable full;
    int waves = 0;
    wgrad_plan(*d, M, full, &waves);
    vf_mlp_bwd_desc sd{};
    WgradTable t{};
    sd.n_fold = d->n_fold;
    int w = 0;
    for (int l = 0; l < d->n_layers; ++l) {
        if (!((layer_mask >> l) & 1u)) continue;
        const int k = sd.n_layers++;
        sd.layer[k] = d->layer[l];
        t.first_wave[k] = w;
        t.rows_per_wave[k] = full.rows_per_wave[l];
        t.part_off[k] = full.part_off[l];
        w += full.first_wave[l + 1] - full.first_wave[l];
    }
    if (sd.n_layers == 0) return VF_OK;
    t.n_layers = sd.n_layers;
    t.first_wave[sd.n_layers] = w;
    if (wgrad_small(*d, M)) hipLaunchKernelGGL((k_mlp_wgrad<true, false>), dim3(w), dim3(64), 0, st, sd, t, partials, M, WgradTail{});
    else hipLaunchKernelGGL((k_mlp_wgrad<false, false>), dim3(w), dim3(64), 0, st, sd, t, partials, M, WgradTail{});
    const int nb = mlp_wgrad_fold_blocks(&sd);
    hipLaunchKernelGGL(k_wgrad_fold, dim3(nb), dim3(kBlock), 0, st, sd, t, (const float*)partials, grad, accumulate, (double*)nullptr, vf_stats_fold{}, nb);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

// waves of k_mlp_wgrad<false, true> the device holds at once (occupancy of THIS kernel x compute units), per device; 0: unknown
static int wgrad_tail_capacity()
{
    static int cap[64];
    static bool known[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!known[dev]) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_mlp_wgrad<false, true>), 64, 0) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
            cap[dev] = per_cu * cus;
        known[dev] = true;
    }
    return cap[dev];
}

// 1: launched (weight gradients + fold + norm + clip + Adam in ONE launch), 0: this layer table / row count / device cannot (the caller
// runs mlp_wgrad_launch + k_adam), < 0: error
int mlp_wgrad_adam_launch(const vf_mlp_bwd_desc* d, float* partials, float* grad, int M, int accumulate, const vf_stats_fold* loss_stats,
                          const vf_wgrad_tail* tl, hipStream_t st)
{
    if (wgrad_small(*d, M)) return fail(0, "fused optimiser tail: streaming row counts (two waves per SIMD) use the separate fold");
    WgradTable t;
    int waves = 0;
    wgrad_plan(*d, M, t, &waves);
    const int cap = wgrad_tail_capacity();
    if (waves > cap) return fail(0, "fused optimiser tail: %d waves do not fit the device at once (%d)", waves, cap);
    WgradTail T{};
    T.grad = grad;
    T.param = tl->param;
    T.m = tl->exp_avg;
    T.v = tl->exp_avg_sq;
    T.sq_part = const_cast<double*>(tl->adam.sumsq_partials);
    T.sync = tl->sync;
    T.adam = tl->adam;
    float bc1, bc2s;
    adam_bias(tl->adam, &bc1, &bc2s);
    T.step = tl->adam.lr / bc1;
    T.bc2_sqrt = bc2s;
    if (loss_stats) T.ls = *loss_stats;
    T.has_ls = loss_stats ? 1 : 0;
    T.n = (int)tl->n;
    T.accumulate = accumulate;
    T.n_fold_blocks = mlp_wgrad_fold_blocks(d);
    static const long long timeout = [] { const char* e = getenv("VISFLY_AMD_FUSED_TAIL_TIMEOUT_MS"); return (long long)(e ? atoi(e) : 2000) * 100000LL; }();
    T.timeout = timeout;        // ticks of the 100 MHz wall clock
#ifdef VF_WGRAD_TRACE
    if (const char* e = getenv("VISFLY_AMD_WGRAD_TRACE_PTR")) T.trace = reinterpret_cast<long long*>(strtoull(e, nullptr, 0));
#endif
    hipLaunchKernelGGL((k_mlp_wgrad<false, true>), dim3(waves), dim3(64), 0, st, *d, t, partials, M, T);
    VF_HIP(hipGetLastError());
    return 1;
}

}  // namespace vf
