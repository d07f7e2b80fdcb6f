// vf_ppo.hip -- PPO inner loop on the device (gfx950): GAE scan, advantage normalisation,
// nn.Linear forward / backward on the fp32 MFMA, squashed-Gaussian head, clipped-surrogate
// loss, gradient clipping + Adam.
//
// Reference: utils/algorithms/PPO.py:177-337 (train), SB3 2.2.1 RolloutBuffer /
// collect_rollouts (mirrored in utils/algorithms/common.py:97-132), utils/policies/policies.py:
// 195-254, utils/policies/extractors.py:376-449.  The reference runs these as torch autograd
// over nn.Linear/ReLU; here each piece is one launch.  GEMMs use v_mfma_f32_32x32x2_f32
// (exact fp32 products, fp32 accumulate, k-ordered), operands staged through LDS with an
// odd row stride (conflict-free ds_read_b32 for the MFMA A/B fragments).
#include <algorithm>
#include <utility>

#include "vf_common.hpp"
#include "vf_adam_device.hpp"
#include "vf_ppo_device.hpp"
#include "vf_env_device.hpp"  // Philox

namespace vf {

using f32x16 = __attribute__((ext_vector_type(16))) float;

#ifdef VF_PROBE   // tools/exp_probe.py: shader-clock time of block 0 per kernel phase (never part of the product build)
__device__ unsigned long long g_probe[32];
#define VF_PROBE_INIT() unsigned long long probe_t = clock64()
#define VF_PROBE_AT(i)                                                \
    do {                                                              \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                    \
            const unsigned long long t_ = clock64();                  \
            g_probe[i] += t_ - probe_t;                               \
            probe_t = t_;                                             \
        }                                                             \
    } while (0)
#else
#define VF_PROBE_INIT()
#define VF_PROBE_AT(i)
#endif

// ------------------------------------------------------------------------------------------------
// GAE: thread per env walks T backwards; loads are coalesced across envs (common.py:119-132)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_gae(const float* __restrict__ r, const float* __restrict__ v,
                                                const float* __restrict__ es, const float* __restrict__ lastv,
                                                const float* __restrict__ dones, float* __restrict__ adv,
                                                float* __restrict__ ret, int T, int N, float gamma, float gl)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    float last = 0.0f;
    float nnt = 1.0f - dones[i];
    float nv = lastv[i];
    for (int t = T - 1; t >= 0; --t) {
        const size_t o = (size_t)t * N + i;
        const float vt = v[o];
        const float delta = r[o] + gamma * nv * nnt - vt;
        last = delta + gl * nnt * last;
        adv[o] = last;
        ret[o] = last + vt;
        nnt = 1.0f - es[o];  // for step t-1: next_non_terminal = 1 - episode_starts[t]
        nv = vt;
    }
}

// TD-lambda returns (utils/algorithms/common.py:893-923): same access pattern as GAE
__global__ __launch_bounds__(kBlock) void k_td_returns(const float* __restrict__ r, const unsigned char* __restrict__ done,
                                                       const unsigned char* __restrict__ ep_done,
                                                       const float* __restrict__ nv, float* __restrict__ ret, int H, int N,
                                                       float gamma, float lamda, float lg, float oml)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    float Ai = 0.0f, lam = 1.0f;
    float Bi = nv[(size_t)(H - 1) * N + i] * (done[(size_t)(H - 1) * N + i] ? 0.0f : 1.0f);
    for (int t = H - 1; t >= 0; --t) {
        const size_t o = (size_t)t * N + i;
        const float active = done[o] ? 0.0f : 1.0f, dm = done[o] ? 1.0f : 0.0f, ea = ep_done[o] ? 0.0f : 1.0f;
        lam = lam * lamda * active + dm;
        Ai = active * ((lg * Ai + gamma * nv[o]) + ((1.0f - lam) / oml) * r[o]);
        Bi = gamma * (nv[o] * dm * ea + Bi * active) + r[o];
        ret[o] = oml * Ai + lam * Bi;
    }
}

// ------------------------------------------------------------------------------------------------
// reductions
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    return x;
}
__device__ __forceinline__ float wave_sumf(float x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    return x;
}

// sum and sum of squares in fp64: per-block partials, then one block folds them (deterministic)
__global__ __launch_bounds__(kBlock) void k_sum2_partial(const float* __restrict__ x, long n, double* __restrict__ part)
{
    __shared__ double sh[2][4];
    double s = 0.0, ss = 0.0;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        const double a = x[i];
        s += a;
        ss += a * a;
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = s; sh[1][w] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        part[2 * blockIdx.x + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
}

__global__ void k_sum2_final(const double* __restrict__ part, int nblk, double* __restrict__ out2, float* out_ss_f32)
{
    // one wave: lane-strided partial sums, then the fixed shuffle tree (deterministic)
    double s = 0.0, ss = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) { s += part[2 * b]; ss += part[2 * b + 1]; }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if (threadIdx.x == 0) {
        if (out2) { out2[0] = s; out2[1] = ss; }
        if (out_ss_f32) *out_ss_f32 = (float)ss;
    }
}

// (A - mean) / (std_unbiased + 1e-8)   (PPO.py:217-220)
__global__ __launch_bounds__(kBlock) void k_adv_apply(const float* __restrict__ a, float* __restrict__ out, long n,
                                                      const double* __restrict__ sums, double count)
{
    const double mean = sums[0] / count;
    double var = (sums[1] - sums[0] * mean) / (count - 1.0);
    var = var > 0.0 ? var : 0.0;
    const float m = (float)mean, sd = (float)sqrt(var) + 1e-8f;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) out[i] = (a[i] - m) / sd;
}

// segmented variant: one block per minibatch (contiguous slice of the shuffled epoch buffer)
__global__ __launch_bounds__(kBlock) void k_adv_seg_sums(const float* __restrict__ x, long seg_len, double* __restrict__ sums)
{
    __shared__ double sh[2][4];
    const float* xs = x + (size_t)blockIdx.x * seg_len;
    double s = 0.0, ss = 0.0;
    for (long i = threadIdx.x; i < seg_len; i += kBlock) {
        const double a = xs[i];
        s += a;
        ss += a * a;
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = s; sh[1][w] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sums[2 * blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        sums[2 * blockIdx.x + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
}

__global__ __launch_bounds__(kBlock) void k_adv_seg_apply(const float* __restrict__ a, float* __restrict__ out, long seg_len,
                                                          const double* __restrict__ sums, double count)
{
    const double s0 = sums[2 * blockIdx.y], s1 = sums[2 * blockIdx.y + 1];
    const double mean = s0 / count;
    double var = (s1 - s0 * mean) / (count - 1.0);
    var = var > 0.0 ? var : 0.0;
    const float m = (float)mean, sd = (float)sqrt(var) + 1e-8f;
    const size_t base = (size_t)blockIdx.y * seg_len;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < seg_len; i += (long)gridDim.x * kBlock)
        out[base + i] = (a[base + i] - m) / sd;
}

// ------------------------------------------------------------------------------------------------
// Linear layers on the fp32 MFMA.  Block = 4 waves, 64 output rows; each wave owns one 32-row
// half and every other 32-column tile.  C/D fragment of v_mfma_f32_32x32x2_f32:
// col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// ------------------------------------------------------------------------------------------------
constexpr int kRows = 64;

// Branch-free MFMA sweeps over a reduction padded to a multiple of 16 (pad columns are zero in LDS):
// per chunk, the fragments of 8 k-pairs are fetched from LDS ahead of the MFMAs that consume them.
// `bs` = LDS stride of one reduction step for the B fragment.  Callers pick the variant with a
// wave-uniform (SGPR) condition, so there is no exec-mask traffic around the matrix instructions.
__device__ __forceinline__ void mfma_sweep1(const float* __restrict__ ap, const float* __restrict__ b0, int bs, int red16,
                                            f32x16& acc0)
{
    for (int k0 = 0; k0 < red16; k0 += 16) {
        float a[8], x0[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = ap[k0 + 2 * j]; x0[j] = b0[(k0 + 2 * j) * bs]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], x0[j], acc0, 0, 0, 0);
    }
}
__device__ __forceinline__ void mfma_sweep2(const float* __restrict__ ap, const float* __restrict__ b0,
                                            const float* __restrict__ b1, int bs, int red16, f32x16& acc0, f32x16& acc1)
{
    for (int k0 = 0; k0 < red16; k0 += 16) {
        float a[8], x0[8], x1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = ap[k0 + 2 * j]; x0[j] = b0[(k0 + 2 * j) * bs]; x1[j] = b1[(k0 + 2 * j) * bs]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], x0[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], x1[j], acc1, 0, 0, 0);
        }
    }
}

// Stage one 64-row tile of A (optionally masked by Ym > 0) into LDS rows of odd stride `sa`.
// Vector path: 16-byte global loads when the row length is a multiple of 4 with a power-of-two
// number of float4 per row (K, No in {4, 8, ..., 128}); scalar path otherwise (K = 13, 3).
template <bool MASK, int NT = kBlock>
__device__ __forceinline__ void stage_rows(float* __restrict__ As, int sa, const float* __restrict__ A, int lda,
                                           const float* __restrict__ Ym, int ldym, int m0, int M, int red, int redp,
                                           int nrows = kRows, int act = VF_ACTIVATION_RELU)
{
    const int tid = threadIdx.x;
    const int c4 = red >> 2;
    const bool vec = (red & 3) == 0 && (c4 & (c4 - 1)) == 0 && c4 <= 32 && (lda & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (!MASK || !Ym || ((ldym & 3) == 0 && (reinterpret_cast<uintptr_t>(Ym) & 15) == 0));
    const bool mask = MASK && Ym != nullptr;
    // wave-uniform 64-bit bases + 32-bit lane offsets (one VGPR per address).  Rows past the matrix read row
    // M-1 again (valid address, no divergent branch around the load) and are zeroed by a select.
    const float* Ab = A + (size_t)m0 * lda;
    const float* Yb = mask ? Ym + (size_t)m0 * ldym : nullptr;
    const int rmax = M - 1 - m0;                  // last valid row of this tile
    if (vec) {
        const int sh = 31 - __clz(c4);            // log2(float4 per row)
        const int col = (tid & (c4 - 1)) << 2, r0 = tid >> sh, rstep = NT >> sh;
        for (int rb0 = 0; rb0 < nrows; rb0 += 8 * rstep) {   // batches of <= 8 independent 16-byte loads, then the LDS writes
            const int rb = rb0 + r0;
            const int nj = min(8, (nrows - rb0 + rstep - 1) / rstep);   // wave-uniform trip count: no loads for rows that do not exist
            float4 v[8], y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < nj) {
                    const int r = min(rb + j * rstep, rmax);
                    v[j] = *reinterpret_cast<const float4*>(Ab + (unsigned)(r * lda + col));
                    if (mask) y[j] = *reinterpret_cast<const float4*>(Yb + (unsigned)(r * ldym + col));
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < nj) {
                    const int r = rb + j * rstep;
                    const bool ok = r <= rmax;
                    float4 x = v[j];
                    if (mask) {
                        x.x = act_mul(x.x, y[j].x, act); x.y = act_mul(x.y, y[j].y, act);
                        x.z = act_mul(x.z, y[j].z, act); x.w = act_mul(x.w, y[j].w, act);
                    }
                    if (r < nrows) {
                        float* d = As + r * sa + col;
                        d[0] = ok ? x.x : 0.0f; d[1] = ok ? x.y : 0.0f; d[2] = ok ? x.z : 0.0f; d[3] = ok ? x.w : 0.0f;
                    }
                }
            }
        }
#ifndef VF_TEST_NO_PAD_ZERO
        if (redp > red) {      // a vector-loadable row that is not a whole MFMA chunk (4- or 8-wide: the action columns of a
            const int np = redp - red;                     // critic): the pad columns must be zeros, not what LDS held before
            for (int i = tid; i < nrows * np; i += NT) As[(i / np) * sa + red + (i % np)] = 0.0f;
        }
#endif
    } else {
        const int total = nrows * redp;
        for (int base = tid; base < total; base += 4 * NT) {     // 4 independent loads in flight per thread
            float x[4], y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = min(base + j * NT, total - 1);
                const int r = idx / redp, k = idx - r * redp;
                const int rc = min(r, rmax), kc = min(k, red - 1);
                x[j] = Ab[(unsigned)(rc * lda + kc)];
                if (mask) y[j] = Yb[(unsigned)(rc * ldym + kc)];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = base + j * NT;
                if (idx < total) {
                    const int r = idx / redp, k = idx - r * redp;
                    float v = x[j];
                    if (mask) v = act_mul(v, y[j], act);
                    As[r * sa + k] = (r <= rmax && k < red) ? v : 0.0f;
                }
            }
        }
    }
}

// Split staging for latency overlap: `load` issues up to 8 independent 16-byte global loads per thread
// into registers (64 rows x <= 128 floats), `store` parks them in LDS later.  Same vector conditions
// as stage_rows; `ok()` false -> the caller falls back to stage_rows.
template <bool MASK>
struct RowPrefetch {
    float4 v[8];
    int sh, col, r0, rstep;
    bool vec;
    __device__ __forceinline__ void setup(const float* A, int lda, const float* Ym, int ldym, int red)
    {
        const int c4 = red >> 2;
        vec = (red & 3) == 0 && (c4 & (c4 - 1)) == 0 && c4 >= 1 && c4 <= 32 && (lda & 3) == 0 &&
              ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
              (!MASK || !Ym || ((ldym & 3) == 0 && (reinterpret_cast<uintptr_t>(Ym) & 15) == 0));
        sh = 31 - __clz(c4 > 0 ? c4 : 1);
        col = (threadIdx.x & (c4 - 1)) << 2;
        r0 = threadIdx.x >> sh;
        rstep = kBlock >> sh;          // rows covered per pass; 64 rows -> 64 / rstep <= 8 passes
    }
    __device__ __forceinline__ void load(const float* __restrict__ A, int lda, const float* __restrict__ Ym, int ldym, int m0,
                                         int M, int act = VF_ACTIVATION_RELU)
    {
        const bool mask = MASK && Ym != nullptr;
        float4 y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {       // rows past the matrix re-read row M-1 (no branch around the load), zeroed below
            const int m = min(m0 + r0 + j * rstep, M - 1);
            v[j] = *reinterpret_cast<const float4*>(A + (size_t)m * lda + col);
            if (mask) y[j] = *reinterpret_cast<const float4*>(Ym + (size_t)m * ldym + col);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = m0 + r0 + j * rstep < M;
            float4 x = v[j];
            if (mask) {
                x.x = act_mul(x.x, y[j].x, act); x.y = act_mul(x.y, y[j].y, act);
                x.z = act_mul(x.z, y[j].z, act); x.w = act_mul(x.w, y[j].w, act);
            }
            v[j] = make_float4(ok ? x.x : 0.0f, ok ? x.y : 0.0f, ok ? x.z : 0.0f, ok ? x.w : 0.0f);
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ As, int sa) const
    {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = r0 + j * rstep;
            if (r < kRows) {
                float* d = As + r * sa + col;
                d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w;
            }
        }
    }
};

// BWD == false: C[m][n] = act(sum_k A[m][k] * W[n][k] + b[n])          (forward; red = K, cols = No)
// BWD == true : C[m][k] = sum_n (A[m][n] * [Ymask[m][n] > 0]) * W[n][k]  (data grad; red = No, cols = K)
// Weight-stationary: a block stages W once and walks 64-row tiles with stride gridDim.x; the next
// tile's rows are fetched into registers while the MFMAs of the current one run.
template <bool BWD>
__global__ __launch_bounds__(kBlock) void k_linear(const float* __restrict__ A, int lda, const float* __restrict__ Ym,
                                                   int ldym, const float* __restrict__ W, const float* __restrict__ bias,
                                                   float* __restrict__ C, int ldc, int M, int K, int No, int accumulate, int act)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int red = BWD ? No : K;         // reduction length
    const int cols = BWD ? K : No;        // output columns
    const int red16 = (red + 15) & ~15;   // the MFMA sweeps consume 16 reduction steps per chunk; pads are zero
    const int ct = (cols + 31) >> 5;      // 32-column tiles
    const int sa = red16 + 1;             // odd LDS row strides
    float* As = lds;                      // [64][sa]
    float* Ws = lds + kRows * sa;         // fwd: [ct*32][sa] (col-major over red) ; bwd: [red16][ct*32+1]
    const int sw = BWD ? ct * 32 + 1 : sa;
    const int tid = threadIdx.x;
    const int ntiles = (M + kRows - 1) / kRows;

    RowPrefetch<BWD> pf;
    pf.setup(A, lda, Ym, ldym, red);
    int tile = blockIdx.x;
    if (pf.vec && tile < ntiles) pf.load(A, lda, Ym, ldym, tile * kRows, M, act);   // in flight while W is staged

    // W image: Ws[n * sw + k] = W[n][k] for both directions (forward reads it column-major over the
    // reduction, the data gradient row-major); pads are zero.
    const int wrows = BWD ? red16 : ct * 32, wcols = BWD ? ct * 32 : red16;
    const bool padded = wrows != No || wcols != K || red16 != red;
    if (padded) {
        for (int idx = tid; idx < kRows * sa + wrows * sw; idx += kBlock) lds[idx] = 0.0f;
        __syncthreads();
    }
    stage_rows<false>(Ws, sw, W, K, nullptr, 0, 0, No, K, K, wrows);

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform on purpose (SGPR control flow)
    const int lane = tid & 63, lr = lane & 31, lk = lane >> 5;
    const int rt = wave & 1;              // row half
    const int c0 = wave >> 1;             // column tiles c0, c0 + 2
    const int nacc = c0 + 2 < ct ? 2 : (c0 < ct ? 1 : 0);
    const float* ap = As + (rt * 32 + lr) * sa + lk;
    const float* b0 = BWD ? Ws + lk * sw + c0 * 32 + lr : Ws + (c0 * 32 + lr) * sw + lk;
    const float* b1 = BWD ? Ws + lk * sw + (c0 + 2) * 32 + lr : Ws + ((c0 + 2) * 32 + lr) * sw + lk;
    const int bs = BWD ? sw : 1;
    for (; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * kRows;
        if (pf.vec) pf.store(As, sa);
        else stage_rows<BWD>(As, sa, A, lda, Ym, ldym, m0, M, red, red, kRows, act);
        __syncthreads();
        if (pf.vec && tile + gridDim.x < ntiles) pf.load(A, lda, Ym, ldym, (tile + gridDim.x) * kRows, M, act);
        f32x16 acc0 = {0}, acc1 = {0};
        if (nacc == 2) mfma_sweep2(ap, b0, b1, bs, red16, acc0, acc1);
        else if (nacc == 1) mfma_sweep1(ap, b0, bs, red16, acc0);
        auto emit = [&](const f32x16& acc, int ctile) {
            const int n = ctile * 32 + lr;
            if (n >= cols) return;
            const float bn = (!BWD && bias) ? bias[n] : 0.0f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lk;
                if (m >= M) continue;
                float y = acc[reg] + bn;
                if (!BWD) y = act_fwd(y, act);
                float* dst = C + (size_t)m * ldc + n;
                *dst = accumulate ? *dst + y : y;
            }
        };
        if (nacc >= 1) emit(acc0, c0);
        if (nacc == 2) emit(acc1, c0 + 2);
        __syncthreads();                  // every wave is done reading As before the next tile overwrites it
    }
}

// ------------------------------------------------------------------------------------------------
// Whole-network forward: activations stay in LDS, weights streamed per layer (see vf_mlp_desc)
// ------------------------------------------------------------------------------------------------
struct MlpIo {
    const float* in[4];
    float* out[2];
};

// MFMA sweeps with the B operand read straight from global memory (L1/L2-resident packed weights,
// coalesced: lane lr = output column) and the A operand from LDS.  Fragments of the next 16 reduction
// steps are fetched while the MFMAs of the current 16 run; the first B chunk is passed in by the caller,
// who issues it before the barrier that publishes the A tile.  `ldb` = floats between reduction steps of B.
struct BFrag {
    float x[8];
};
// bg = wave-uniform base of the layer's packed image, off = this lane's float offset for reduction step 0
__device__ __forceinline__ BFrag load_bfrag(const float* __restrict__ bg, int off, int ldb)
{
    BFrag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.x[j] = bg[(unsigned)(off + 2 * j * ldb)];
    return f;
}
__device__ __forceinline__ void mfma_sweep_gb1(const float* __restrict__ ap, const float* __restrict__ bg, int off0, int ldb,
                                               int red16, BFrag x0, f32x16& acc0)
{
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = ap[2 * j];
    for (int k0 = 0; k0 < red16; k0 += 16) {
        float an[8];
        BFrag n0;
        if (k0 + 16 < red16) {
            n0 = load_bfrag(bg, off0 + (k0 + 16) * ldb, ldb);
#pragma unroll
            for (int j = 0; j < 8; ++j) an[j] = ap[k0 + 16 + 2 * j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], x0.x[j], acc0, 0, 0, 0);
        if (k0 + 16 < red16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[j] = an[j]; x0.x[j] = n0.x[j]; }
        }
    }
}
__device__ __forceinline__ void mfma_sweep_gb2(const float* __restrict__ ap, const float* __restrict__ bg, int off0, int off1,
                                               int ldb, int red16, BFrag x0, BFrag x1, f32x16& acc0, f32x16& acc1)
{
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = ap[2 * j];
    for (int k0 = 0; k0 < red16; k0 += 16) {
        float an[8];
        BFrag n0, n1;
        if (k0 + 16 < red16) {
            n0 = load_bfrag(bg, off0 + (k0 + 16) * ldb, ldb);
            n1 = load_bfrag(bg, off1 + (k0 + 16) * ldb, ldb);
#pragma unroll
            for (int j = 0; j < 8; ++j) an[j] = ap[k0 + 16 + 2 * j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], x0.x[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], x1.x[j], acc1, 0, 0, 0);
        }
        if (k0 + 16 < red16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[j] = an[j]; x0.x[j] = n0.x[j]; x1.x[j] = n1.x[j]; }
        }
    }
}

// Fully unrolled sweep for a compile-time chunk count (NCH x 16 reduction steps, NACC accumulators): three chunks of B
// fragments are in flight (every launch starts with cold L2s and all CUs walk the layers in lock-step, so each weight
// chunk is a first-touch miss of ~2 k cycles), buffers rotate by NAME -- no register copies, no branches -- so that
// hipcc's waitcnt insertion can leave the younger chunks outstanding (vmcnt(N) instead of vmcnt(0)).
template <int NCH, int NACC>
__device__ __forceinline__ void mfma_sweep_static(const float* __restrict__ ap, const float* __restrict__ bg, int off0, int off1,
                                                  int ldb, f32x16& acc0, f32x16& acc1)
{
    constexpr int D = NCH < 3 ? NCH : 3;
    float xb[3][2][8];
    float a[2][8];
#pragma unroll
    for (int c = 0; c < D; ++c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            xb[c][0][j] = bg[(unsigned)(off0 + (16 * c + 2 * j) * ldb)];
            if (NACC == 2) xb[c][1][j] = bg[(unsigned)(off1 + (16 * c + 2 * j) * ldb)];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) a[0][j] = ap[2 * j];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        // load group: A fragments of the next chunk (LDS) and the refill of the B buffer chunk c-1 just released
        if (c + 1 < NCH) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[(c + 1) & 1][j] = ap[16 * (c + 1) + 2 * j];
        }
        if (c >= 1 && c - 1 + D < NCH) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xb[(c - 1) % 3][0][j] = bg[(unsigned)(off0 + (16 * (c - 1 + D) + 2 * j) * ldb)];
                if (NACC == 2) xb[(c - 1) % 3][1][j] = bg[(unsigned)(off1 + (16 * (c - 1 + D) + 2 * j) * ldb)];
            }
        }
        // MFMA group, nothing in between: any other instruction between two MFMAs on the same accumulator costs
        // ~43 extra cycles (MI355X_MICROARCH.md, per-instruction constants), and the narrow layers have one accumulator
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {   // touch every operand of the group: ONE s_waitcnt in front instead of one per MFMA
            asm volatile("" : "+v"(a[c & 1][j]), "+v"(xb[c % 3][0][j]));
            if (NACC == 2) asm volatile("" : "+v"(xb[c % 3][1][j]));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c & 1][j], xb[c % 3][0][j], acc0, 0, 0, 0);
            if (NACC == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c & 1][j], xb[c % 3][1][j], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Packed forward weights: per layer Wt[k][n] = W[n][k] for k < K16 = round16(K), n < N32 = round32(No), zero padded,
// at float offset wt_off of the packed buffer (vf_mlp_pack_weights) -- the forward B operand without any guard.
__global__ __launch_bounds__(kBlock) void k_mlp_pack_weights(const vf_mlp_desc d, const float* __restrict__ params,
                                                             float* __restrict__ packed)
{
    const vf_mlp_layer L = d.layer[blockIdx.y];
    if ((int)blockIdx.y >= d.n_layers) return;
    const int K16 = (L.K + 15) & ~15, N32 = (L.No + 31) & ~31;
    for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < K16 * N32; idx += gridDim.x * kBlock) {
        const int k = idx / N32, n = idx - k * N32;
        packed[L.wt_off + idx] = (k < L.K && n < L.No) ? params[L.w_off + n * L.K + k] : 0.0f;
    }
    // data-gradient image: Wb[n][k] = W[n][k] for n < round16(No), k < round32(K), zero padded
    const int N16 = (L.No + 15) & ~15, K32 = (L.K + 31) & ~31;
    for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < N16 * K32; idx += gridDim.x * kBlock) {
        const int n = idx / K32, k = idx - n * K32;
        packed[L.wb_off + idx] = (k < L.K && n < L.No) ? params[L.w_off + n * L.K + k] : 0.0f;
    }
    // register-chain image (vf_mlp_chain.hip): block (a, g) = A fragments of four reduction steps, float4 per lane
    const bool nat = L.src < 4;                       // reads an observation: natural k order
    const int G = nat ? (L.K + 7) >> 3 : ((L.K + 31) >> 5) * 4, NT = (L.No + 31) >> 5;
    for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < NT * G * 256; idx += gridDim.x * kBlock) {
        const int j = idx & 3, l = (idx >> 2) & 63, blk = idx >> 8, a = blk / G, g = blk - a * G;
        const int n = 32 * a + (l & 31), h = l >> 5;
        const int k = nat ? 8 * g + 2 * j + h : 32 * (g >> 2) + 8 * (g & 3) + 4 * h + j;
        packed[L.wr_off + idx] = (k < L.K && n < L.No) ? params[L.w_off + n * L.K + k] : 0.0f;
    }
    // reverse-chain image: block (a, g), a < ceil(K / 32), g < ceil(No / 8): W[32 (g / 4) + 8 (g % 4) + 4 h + j][32 a + (l & 31)]
    const int GQ = (L.No + 7) >> 3, KT = (L.K + 31) >> 5;
    for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < KT * GQ * 256; idx += gridDim.x * kBlock) {
        const int j = idx & 3, l = (idx >> 2) & 63, blk = idx >> 8, a = blk / GQ, g = blk - a * GQ;
        const int k = 32 * a + (l & 31), n = 32 * (g >> 2) + 8 * (g & 3) + 4 * (l >> 5) + j;
        packed[L.wq_off + idx] = (k < L.K && n < L.No) ? params[L.w_off + n * L.K + k] : 0.0f;
    }
}

__global__ __launch_bounds__(kBlock, 2) void k_mlp_forward(const vf_mlp_desc d, const float* __restrict__ params,
                                                           const float* __restrict__ packed, const MlpIo io, int M)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 31, lk = lane >> 5;
    const int rt = wave & 1, c0 = wave >> 1;
    const int ntiles = (M + kRows - 1) / kRows;
    VF_PROBE_INIT();
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * kRows;
        const bool full = m0 + kRows <= M;                 // no row guards in the epilogue except on the last tile
        __syncthreads();                                   // previous tile is completely consumed
        VF_PROBE_AT(0);
        for (int b = 0; b < d.n_inputs; ++b) {              // observations -> LDS
            const int w = d.in_dim[b], wp16 = (w + 15) & ~15;   // zero padded to the MFMA chunk
            stage_rows<false>(lds + d.lds_off[b], d.lds_stride[b], io.in[b], w, nullptr, 0, m0, M, w, wp16);
        }
        VF_PROBE_AT(1);
        for (int li = 0; li < d.n_layers; ++li) {
            const vf_mlp_layer L = d.layer[li];
            const int red16 = (L.K + 15) & ~15, ct = (L.No + 31) >> 5, ldb = ct * 32;
            const int nacc = c0 + 2 < ct ? 2 : (c0 < ct ? 1 : 0);
            const float* bg = packed + L.wt_off;            // wave-uniform base, 32-bit lane offsets
            const int off0 = lk * ldb + c0 * 32 + lr, off1 = off0 + 64;
#ifndef VF_TEST_NO_PAD_ZERO                                 // (tools/exp_pad_poison.py: shows that the poison test fails without)
            if (L.src >= 4 && (L.K & 15)) {
                // a hidden source whose width is not a multiple of the 16-step MFMA chunk (a concatenation such as
                // features (+) action = 68 columns): the sweep reads [K, round16(K)) as well.  The packed weights are zero
                // there, but 0 x (whatever the recycled LDS region holds) is NaN for NaN / Inf bit patterns, which the ReLU
                // then turns into a silent 0.  Nobody else writes those columns: zero them (rows 64 x < 16 columns).
                float* pad = lds + d.lds_off[L.src] + L.src_col + L.K;
                const int np = red16 - L.K, ss = d.lds_stride[L.src];
                for (int i = tid; i < kRows * np; i += kBlock) pad[(i / np) * ss + (i % np)] = 0.0f;
            }
#endif
            __syncthreads();                               // inputs of this layer are in LDS
            VF_PROBE_AT(2);
            const float* As = lds + d.lds_off[L.src] + L.src_col;
            const int sa = d.lds_stride[L.src];
            const float* ap = As + (rt * 32 + lr) * sa + lk;
            f32x16 acc0 = {0}, acc1 = {0};
            const int nch = red16 >> 4;                    // 1 (K = 13, 3), 4 (K = 64), 8 (K = 128): unrolled sweeps; else generic
            if (nacc == 2) {
                if (nch == 8) mfma_sweep_static<8, 2>(ap, bg, off0, off1, ldb, acc0, acc1);
                else if (nch == 4) mfma_sweep_static<4, 2>(ap, bg, off0, off1, ldb, acc0, acc1);
                else if (nch == 1) mfma_sweep_static<1, 2>(ap, bg, off0, off1, ldb, acc0, acc1);
                else mfma_sweep_gb2(ap, bg, off0, off1, ldb, red16, load_bfrag(bg, off0, ldb), load_bfrag(bg, off1, ldb), acc0, acc1);
            } else if (nacc == 1) {
                if (nch == 8) mfma_sweep_static<8, 1>(ap, bg, off0, off1, ldb, acc0, acc1);
                else if (nch == 4) mfma_sweep_static<4, 1>(ap, bg, off0, off1, ldb, acc0, acc1);
                else if (nch == 1) mfma_sweep_static<1, 1>(ap, bg, off0, off1, ldb, acc0, acc1);
                else mfma_sweep_gb1(ap, bg, off0, ldb, red16, load_bfrag(bg, off0, ldb), acc0);
            }
            VF_PROBE_AT(5);
            // epilogue: bias + ReLU, into the destination region (LDS or global) and the optional saved copy
            const int rb = rt * 32 + 4 * lk;               // first row of this lane's accumulator column
            auto emit = [&](f32x16 acc, int ctile) {
                const int n = ctile * 32 + lr;
                if (n >= L.No) return;
                const float bn = params[L.b_off + n];
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    float y = acc[reg] + bn;
                    y = act_fwd(y, L.relu);
                    acc[reg] = y;
                }
                if (L.dst < VF_MLP_OUT0) {
                    float* dl = lds + d.lds_off[L.dst] + L.dst_col + rb * d.lds_stride[L.dst] + n;
                    const int sd = d.lds_stride[L.dst];
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) dl[((reg & 3) + 8 * (reg >> 2)) * sd] = acc[reg];
                }
                float* gp = nullptr;                         // wave-uniform base of this tile's rows
                int ldg = 0;
                if (L.dst >= VF_MLP_OUT0) { gp = io.out[L.dst - VF_MLP_OUT0] + (size_t)m0 * L.No; ldg = L.No; }
                else if (L.save) { gp = L.save + (size_t)m0 * L.save_ld + L.dst_col; ldg = L.save_ld; }
                if (gp) {
                    const int o = rb * ldg + n;
                    if (full) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) gp[(unsigned)(o + ((reg & 3) + 8 * (reg >> 2)) * ldg)] = acc[reg];
                    } else {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int ro = (reg & 3) + 8 * (reg >> 2);
                            if (m0 + rb + ro < M) gp[(unsigned)(o + ro * ldg)] = acc[reg];
                        }
                    }
                }
            };
            if (nacc >= 1) emit(acc0, c0);
            if (nacc == 2) emit(acc1, c0 + 2);
            VF_PROBE_AT(6);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Whole-network backward: block-private row tiles, layer-major sweep (see vf_mlp_bwd_desc)
// ------------------------------------------------------------------------------------------------
constexpr int kBwdThreads = 512;   // 8 waves: (row half) x (4 column tiles) for the data gradient, 2 dW tiles each

// Register prefetch of one [64][w] fp32 tile by 512 threads: `issue` starts the global loads one work item
// ahead, `park` writes them to LDS rows of stride `sa` (optionally masked by a second prefetched tile > 0,
// rows past the matrix and the pad columns w..wpad zeroed).  mode 1: 16-byte loads (w/4 a power of two,
// <= 4 per thread); mode 2: narrow tiles (w <= 16, 2 scalars per thread); mode 0: staged directly at park time.
struct TilePf {
    float4 v[4];
    static __device__ __forceinline__ int mode_of(const float* A, int lda, int w)
    {
        const int c4 = w >> 2;
        if ((w & 3) == 0 && c4 >= 1 && (c4 & (c4 - 1)) == 0 && c4 <= 32 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0) return 1;
        return w <= 16 ? 2 : 0;
    }
    __device__ __forceinline__ void issue(int mode, const float* __restrict__ A, int lda, int m0, int M, int w)
    {
        const int tid = threadIdx.x, rmax = M - 1 - m0;
        const float* Ab = A + (size_t)m0 * lda;               // wave-uniform base, 32-bit lane offsets
        if (mode == 1) {
            const int c4 = w >> 2, sh = 31 - __clz(c4), col = (tid & (c4 - 1)) << 2, r0 = tid >> sh, rstep = kBwdThreads >> sh;
            const int nj = rstep >= kRows ? 1 : kRows / rstep;   // 4, 2 or 1 rows per thread
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nj) v[j] = *reinterpret_cast<const float4*>(Ab + (unsigned)(min(r0 + j * rstep, rmax) * lda + col));
        } else if (mode == 2) {
            const int total = kRows * w;
            float t[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int idx = min(tid + j * kBwdThreads, total - 1), r = idx / w, k = idx - r * w;
                t[j] = Ab[(unsigned)(min(r, rmax) * lda + k)];
            }
            v[0].x = t[0]; v[0].y = t[1];
        }
    }
    __device__ __forceinline__ void park(int mode, float* __restrict__ As, int sa, int m0, int M, int w, int wpad, const TilePf* ym,
                                         int act = VF_ACTIVATION_RELU) const
    {
        const int tid = threadIdx.x, rmax = M - 1 - m0;
        if (wpad > w) {                                        // pad columns hold stale words of another layer
            const int pw = wpad - w;
            for (int idx = tid; idx < kRows * pw; idx += kBwdThreads) {
                const int r = idx / pw, k = w + idx - r * pw;
                As[r * sa + k] = 0.0f;
            }
        }
        if (mode == 1) {
            const int c4 = w >> 2, sh = 31 - __clz(c4), col = (tid & (c4 - 1)) << 2, r0 = tid >> sh, rstep = kBwdThreads >> sh;
            const int nj = rstep >= kRows ? 1 : kRows / rstep;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < nj) {
                    const int r = r0 + j * rstep;
                    float4 x = v[j];
                    if (ym) {
                        const float4 y = ym->v[j];
                        x.x = act_mul(x.x, y.x, act); x.y = act_mul(x.y, y.y, act);
                        x.z = act_mul(x.z, y.z, act); x.w = act_mul(x.w, y.w, act);
                    }
                    const bool ok = r <= rmax;
                    if (r < kRows) {
                        float* d = As + r * sa + col;
                        d[0] = ok ? x.x : 0.0f; d[1] = ok ? x.y : 0.0f; d[2] = ok ? x.z : 0.0f; d[3] = ok ? x.w : 0.0f;
                    }
                }
            }
        } else if (mode == 2) {
            const int total = kRows * w;
            const float t[2] = {v[0].x, v[0].y};
            const float ty[2] = {ym ? ym->v[0].x : 1.0f, ym ? ym->v[0].y : 1.0f};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int idx = tid + j * kBwdThreads;
                if (idx < total) {
                    const int r = idx / w, k = idx - r * w;
                    As[r * sa + k] = r <= rmax ? (ym ? act_mul(t[j], ty[j], act) : t[j]) : 0.0f;
                }
            }
        }
    }
};

// Work items of a block: (layer, tile) in layer-major order over the block's own tiles.  While the MFMAs of item
// i run on LDS buffer i&1, the global loads of item i+1 (saved input X, saved output Y for the ReLU mask and --
// when its producer is not item i itself -- the upstream gradient dY) are in flight into registers; they are
// parked in the other buffer behind one barrier.  The data-gradient B operand streams from the packed weights.
__global__ __launch_bounds__(kBwdThreads) void k_mlp_backward(const vf_mlp_bwd_desc d, const float* __restrict__ packed,
                                                            float* __restrict__ part, int M)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 31, lk = lane >> 5;
    const int rt = wave & 1, c0 = wave >> 1;   // c0 = 0..3: this wave's 32-column tile of dX
    const int mtiles = (M + kRows - 1) / kRows;
    const int T = (mtiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this block (>= 1)
    const int nitems = d.n_layers * T;
    const bool early_dy = T >= 2;              // with one tile per block the producer of the next dY is the current item
    constexpr int kBuf = kRows * (129 + 129);  // floats per staging buffer (max strides)
    float* Bs = lds + 2 * kBuf;                // [kBwdThreads] bias partial sums
    float* prow = part + (size_t)blockIdx.x * d.n_fold;
    VF_PROBE_INIT();

    TilePf pfX, pfY, pfD;
    auto geometry = [&](const vf_mlp_bwd_layer& L, int& sd, int& sx) {
        sd = ((L.No + 31) & ~31) + 1;
        sx = ((L.K + 31) & ~31) + 1;
    };
    auto modes = [&](const vf_mlp_bwd_layer& L, int& mx, int& md) {
        mx = TilePf::mode_of(L.X, L.ld_x, L.K);
        md = TilePf::mode_of(L.dY, L.ld_dy, L.No);
        if (L.Y && TilePf::mode_of(L.Y, L.ld_y, L.No) != md) md = 0;   // mask and gradient must share the thread mapping
    };
    auto park_item = [&](const vf_mlp_bwd_layer& L, int m0, float* buf, bool dy_late) {
        int sd, sx, mx, md;
        geometry(L, sd, sx);
        modes(L, mx, md);
        float* Ds = buf;
        float* Xs = buf + kRows * sd;
        if (dy_late && md) {
            pfD.issue(md, L.dY, L.ld_dy, m0, M, L.No);
        }
        if (mx) pfX.park(mx, Xs, sx, m0, M, L.K, sx - 1, nullptr);
        else {
            if (sx - 1 > L.K) pfX.park(0, Xs, sx, m0, M, L.K, sx - 1, nullptr);   // pads only
            stage_rows<false, kBwdThreads>(Xs, sx, L.X, L.ld_x, nullptr, 0, m0, M, L.K, L.K);
        }
        if (md) pfD.park(md, Ds, sd, m0, M, L.No, sd - 1, L.Y ? &pfY : nullptr, L.act);
        else {
            if (sd - 1 > L.No) pfD.park(0, Ds, sd, m0, M, L.No, sd - 1, nullptr);
            stage_rows<true, kBwdThreads>(Ds, sd, L.dY, L.ld_dy, L.Y, L.ld_y, m0, M, L.No, L.No, kRows, L.act);
        }
    };
    auto issue_item = [&](const vf_mlp_bwd_layer& L, int m0, bool with_dy) {
        int mx, md;
        modes(L, mx, md);
        if (mx) pfX.issue(mx, L.X, L.ld_x, m0, M, L.K);
        if (md && L.Y) pfY.issue(md, L.Y, L.ld_y, m0, M, L.No);
        if (md && with_dy) pfD.issue(md, L.dY, L.ld_dy, m0, M, L.No);
    };

    {   // item 0
        const int m0 = (int)blockIdx.x * kRows;
        issue_item(d.layer[0], m0, true);
        park_item(d.layer[0], m0, lds, false);
    }
    __syncthreads();
    VF_PROBE_AT(8);

    f32x16 acc[2] = {{0}, {0}};
    float bsum = 0.0f;
    int li = 0, ti = 0;                        // layer / tile index of the current item
    for (int it = 0; it < nitems; ++it) {
        const vf_mlp_bwd_layer L = d.layer[li];
        const int K = L.K, No = L.No;
        const int nt = (No + 31) >> 5, kt = (K + 31) >> 5;
        const int sd = nt * 32 + 1, sx = kt * 32 + 1, ldb = kt * 32, red16 = (No + 15) & ~15;
        float* buf = lds + (it & 1) * kBuf;
        const float* Ds = buf;
        const float* Xs = buf + kRows * sd;
        const int m0 = ((int)blockIdx.x + ti * (int)gridDim.x) * kRows;
        const int nli = ti + 1 == T ? li + 1 : li, nti = ti + 1 == T ? 0 : ti + 1;   // next item
        const bool has_next = it + 1 < nitems;
        const int nm0 = ((int)blockIdx.x + nti * (int)gridDim.x) * kRows;
        if (has_next) issue_item(d.layer[nli], nm0, early_dy);
        const int nacc = c0 < kt ? 1 : 0;
        const float* bg = packed + L.wb_off;   // wave-uniform base, 32-bit lane offsets
        const int off0 = lk * ldb + c0 * 32 + lr;
        BFrag x0;
        if (L.need_dx && nacc) x0 = load_bfrag(bg, off0, ldb);
        VF_PROBE_AT(9);
        const int cgrp = No <= 64 ? 64 : 128;
        {   // bias gradient: thread = (column, row slice)
            const int c = tid & (cgrp - 1), sl = tid / cgrp, rows = kRows * cgrp / kBwdThreads;
            if (c < No) {
                float s0 = 0.0f, s1 = 0.0f;
                const float* dp = Ds + (sl * rows) * sd + c;
                for (int r = 0; r < rows; r += 2) { s0 += dp[r * sd]; s1 += dp[(r + 1) * sd]; }
                bsum += s0 + s1;
            }
        }
        const int wtiles = nt * kt;            // <= 16 weight-gradient tiles of 32x32; wave takes wave, wave+8
#pragma unroll
        for (int q = 0; q < 2; ++q) {          // dW[n][k] += sum_m dYm[m][n] X[m][k]
            const int wt = wave + 8 * q;
            if (wt >= wtiles) break;
            const int itn = wt / kt, jt = wt - itn * kt;
            const float* ap = Ds + lk * sd + itn * 32 + lr;
            const float* bp = Xs + lk * sx + jt * 32 + lr;
            f32x16 c = acc[q];
#pragma unroll
            for (int k0 = 0; k0 < kRows; k0 += 16) {
                float fa[8], fb[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { fa[j] = ap[(k0 + 2 * j) * sd]; fb[j] = bp[(k0 + 2 * j) * sx]; }
#pragma unroll
                for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[j], c, 0, 0, 0);
            }
            acc[q] = c;
        }
        VF_PROBE_AT(13);
        if (L.need_dx && nacc) {               // dX[m][k] = sum_n dYm[m][n] W[n][k]
            const float* ap = Ds + (rt * 32 + lr) * sd + lk;
            f32x16 a = {0};
            mfma_sweep_gb1(ap, bg, off0, ldb, red16, x0, a);
            float* dxb = L.dX + (size_t)m0 * L.ld_dx;   // wave-uniform base of this tile's rows
            const int n = c0 * 32 + lr;
            if (n < K) {
                const int rb = rt * 32 + 4 * lk, o = rb * L.ld_dx + n, rmax = M - 1 - m0 - rb;
                if (L.need_dx == 2) {          // second consumer of the same activation: add (all loads first)
                    float old[16];
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        old[reg] = dxb[(unsigned)(min(rb + (reg & 3) + 8 * (reg >> 2), M - 1 - m0) * L.ld_dx + n)];   // rows past M: clamped, never stored
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) a[reg] += old[reg];
                }
                if (rmax >= 27) {              // every row of this lane's column exists
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) dxb[(unsigned)(o + ((reg & 3) + 8 * (reg >> 2)) * L.ld_dx)] = a[reg];
                } else {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int ro = (reg & 3) + 8 * (reg >> 2);
                        if (ro <= rmax) dxb[(unsigned)(o + ro * L.ld_dx)] = a[reg];
                    }
                }
            }
        }
        VF_PROBE_AT(14);
        const bool layer_done = ti + 1 == T;
        if (layer_done) {                      // one partial per layer and block: weights, then the bias column sums
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int wt = wave + 8 * q;
                if (wt >= wtiles) break;
                const int itn = wt / kt, jt = wt - itn * kt;
                const int k = jt * 32 + lr;
                if (k < K) {
                    float* pw = prow + L.w_off;    // wave-uniform base
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int n = itn * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lk;
                        if (n < No) pw[(unsigned)(n * K + k)] = acc[q][reg];
                    }
                }
                acc[q] = f32x16{0};
            }
            Bs[tid] = bsum;
            bsum = 0.0f;
        }
        VF_PROBE_AT(10);
        __syncthreads();                       // buffer it&1 is consumed, this item's dX stores have completed
        VF_PROBE_AT(15);
        if (layer_done && tid < No) {
            float t = 0.0f;
            const int nsl = kBwdThreads / cgrp;
            for (int q = 0; q < nsl; ++q) t += Bs[q * cgrp + tid];
            prow[L.b_off + tid] = t;
        }
        if (has_next) park_item(d.layer[nli], nm0, lds + ((it + 1) & 1) * kBuf, !early_dy);
        VF_PROBE_AT(11);
        __syncthreads();
        VF_PROBE_AT(12);
        li = nli; ti = nti;
    }
}

// dW[n][k] = sum_m dYm[m][n] X[m][k]; block = one chunk of rows, partial written to part[blk][No*K + No]
__global__ __launch_bounds__(kBlock) void k_linear_wgrad(const float* __restrict__ dY, int lddy, const float* __restrict__ Ym,
                                                         int ldym, const float* __restrict__ X, int ldx,
                                                         float* __restrict__ part, int M, int K, int No, int rows_per_block, int act)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nt = (No + 31) >> 5, kt = (K + 31) >> 5;
    const int sd = nt * 32 + 1, sx = kt * 32 + 1;
    float* Ds = lds;               // [64][sd]  masked dY rows
    float* Xs = lds + kRows * sd;  // [64][sx]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 31, lk = lane >> 5;
    const int mb = blockIdx.x * rows_per_block;
    const int me = min(M, mb + rows_per_block);
    const int ntiles = nt * kt;  // <= 16, wave takes tiles wave, wave+4, ...
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    float bsum = 0.0f;  // thread tid < No: column sum of masked dY
    for (int idx = tid; idx < kRows * (sd + sx); idx += kBlock) lds[idx] = 0.0f;   // pad columns stay zero
    __syncthreads();
    for (int m0 = mb; m0 < me; m0 += kRows) {
        stage_rows<true>(Ds, sd, dY, lddy, Ym, ldym, m0, me, No, nt * 32, kRows, act);
        stage_rows<false>(Xs, sx, X, ldx, nullptr, 0, m0, me, K, kt * 32);
        __syncthreads();
        {   // bias gradient: every thread owns (column, row-slice); slices are combined at the end
            const int cgrp = No <= 64 ? 64 : 128, c = tid & (cgrp - 1), part = tid / cgrp, rows = kRows * cgrp / kBlock;
            if (c < No) {
                float s0 = 0.0f, s1 = 0.0f;
                const float* dp = Ds + (part * rows) * sd + c;
                for (int r = 0; r < rows; r += 2) { s0 += dp[r * sd]; s1 += dp[(r + 1) * sd]; }
                bsum += s0 + s1;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int tile = wave + 4 * q;
            if (tile >= ntiles) break;
            const int it = tile / kt, jt = tile - it * kt;
            const float* ap = Ds + lk * sd + it * 32 + lr;
            const float* bp = Xs + lk * sx + jt * 32 + lr;
            f32x16 c = acc[q];
#pragma unroll
            for (int k0 = 0; k0 < kRows; k0 += 16) {   // fetch 8 fragment pairs, then 8 MFMAs
                float fa[8], fb[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { fa[j] = ap[(k0 + 2 * j) * sd]; fb[j] = bp[(k0 + 2 * j) * sx]; }
#pragma unroll
                for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[j], c, 0, 0, 0);
            }
            acc[q] = c;
        }
        __syncthreads();
    }
    float* p = part + (size_t)blockIdx.x * ((size_t)No * K + No);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int tile = wave + 4 * q;
        if (tile >= ntiles) break;
        const int it = tile / kt, jt = tile - it * kt;
        const int k = jt * 32 + lr;
        if (k >= K) continue;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int n = it * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lk;
            if (n < No) p[(size_t)n * K + k] = acc[q][reg];
        }
    }
    {
        __syncthreads();
        const int cgrp = No <= 64 ? 64 : 128, c = tid & (cgrp - 1), part = tid / cgrp, nparts = kBlock / cgrp;
        lds[part * cgrp + c] = bsum;
        __syncthreads();
        if (tid < No) {
            float t = 0.0f;
            for (int q = 0; q < nparts; ++q) t += lds[q * cgrp + tid];
            p[(size_t)No * K + tid] = t;
        }
    }
}

// deterministic second stage: a block owns 64 consecutive output elements (one 256-byte row segment per wave-load);
// wave w sums the partial rows b = w, w+4, w+8, ... with four independent chains, the four waves are combined through
// LDS in a fixed order
__global__ __launch_bounds__(kBlock) void k_fold_partials(const float* __restrict__ part, int nblk, int stride, int nw, int nb,
                                                          float* __restrict__ dW, float* __restrict__ db, int accumulate)
{
    const int n = nw + nb;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int e = blockIdx.x * 64 + lane;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (e < n) {
        const float* p = part + e;
        int r = wave;
        for (; r + 12 < nblk; r += 16) {
            s0 += p[(size_t)r * stride];
            s1 += p[(size_t)(r + 4) * stride];
            s2 += p[(size_t)(r + 8) * stride];
            s3 += p[(size_t)(r + 12) * stride];
        }
        for (; r < nblk; r += 4) s0 += p[(size_t)r * stride];
    }
    __shared__ float sh[4][64];
    sh[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && e < n) {
        const float t = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
        if (e < nw) dW[e] = accumulate ? dW[e] + t : t;
        else if (db) db[e - nw] = accumulate ? db[e - nw] + t : t;
    }
}

// ------------------------------------------------------------------------------------------------
// Squashed diagonal Gaussian head
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_head_sample(const float4* __restrict__ mean, const float* __restrict__ log_std,
                                                        float4* __restrict__ action, float* __restrict__ logp, int M,
                                                        unsigned long long seed, unsigned long long step, int deterministic)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= M) return;
    float4 a;
    logp[i] = head_sample_row(mean[i], log_std, i, seed, step, deterministic, a);
    action[i] = a;
}

// PPO clipped surrogate + value MSE + "entropy" (= mean log-prob for the squashed head), PPO.py:210-263
__global__ __launch_bounds__(kBlock) void k_ppo_loss(const float4* __restrict__ mean, const float* __restrict__ value,
                                                     const float* __restrict__ log_std, const float4* __restrict__ action,
                                                     const float* __restrict__ old_lp, const float* __restrict__ adv,
                                                     const float* __restrict__ ret, float4* __restrict__ d_mean,
                                                     float* __restrict__ d_value, float* __restrict__ part, int M,
                                                     const vf_ppo_loss_cfg cfg)
{
    __shared__ float sh[4][kStats];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    float st[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (i < M) {
        const float4 m4 = mean[i], a4 = action[i];
        const float mu[4] = {m4.x, m4.y, m4.z, m4.w}, a[4] = {a4.x, a4.y, a4.z, a4.w};
        const float ls[4] = {log_std[0], log_std[1], log_std[2], log_std[3]};
        float dm[4], dvl;
        ppo_row(mu, value[i], ls, a, old_lp[i], adv[i], ret[i], cfg, dm, dvl, st, i);
        d_mean[i] = make_float4(dm[0], dm[1], dm[2], dm[3]);
        d_value[i] = dvl;
    }
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float s = wave_sumf(st[k]);
        if ((threadIdx.x & 63) == 0) sh[w][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        const int k = threadIdx.x;
        part[(size_t)blockIdx.x * kStats + k] = (sh[0][k] + sh[1][k]) + (sh[2][k] + sh[3][k]);
    }
}

// ---- first-order policy optimisation glue (BPTT.py:107-134): reparameterised action, its reverse, loss bookkeeping ----
// a = tanh(mean + exp(log_std) * eps)   (reparameterised squashed Gaussian, one thread per row)
__global__ __launch_bounds__(kBlock) void k_reparam_fwd(const float4* __restrict__ mean, const float* __restrict__ log_std,
                                                        const float4* __restrict__ eps, float4* __restrict__ action, int N)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const float4 m = mean[i], e = eps[i];
    action[i] = make_float4(tanhf(m.x + expf(log_std[0]) * e.x), tanhf(m.y + expf(log_std[1]) * e.y),
                            tanhf(m.z + expf(log_std[2]) * e.z), tanhf(m.w + expf(log_std[3]) * e.w));
}

// d_mean = d_action * (1 - a^2);  g_log_std += d_mean * exp(log_std) * eps   (per row; summed over rows by the caller)
__global__ __launch_bounds__(kBlock) void k_reparam_bwd(const float4* __restrict__ d_action, const float4* __restrict__ action,
                                                        const float* __restrict__ log_std, const float4* __restrict__ eps,
                                                        float4* __restrict__ d_mean, float4* __restrict__ g_log_std, int N)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const float4 da = d_action[i], a = action[i], e = eps[i];
    const float4 dm = make_float4(da.x * (1.0f - a.x * a.x), da.y * (1.0f - a.y * a.y), da.z * (1.0f - a.z * a.z),
                                  da.w * (1.0f - a.w * a.w));
    d_mean[i] = dm;
    float4 g = g_log_std[i];
    g.x += dm.x * expf(log_std[0]) * e.x; g.y += dm.y * expf(log_std[1]) * e.y;
    g.z += dm.z * expf(log_std[2]) * e.z; g.w += dm.w * expf(log_std[3]) * e.w;
    g_log_std[i] = g;
}

// loss_i += -reward_i * disc_i; d_reward_i = -disc_i * scale; disc_i <- disc_i * gamma * ~done_i + done_i   (BPTT.py:123-124)
__global__ __launch_bounds__(kBlock) void k_bptt_accumulate(const float* __restrict__ reward, const uint8_t* __restrict__ done,
                                                            float* __restrict__ disc, float* __restrict__ loss,
                                                            float* __restrict__ d_reward, float gamma, float scale, int N)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const float d = disc[i];
    loss[i] = loss[i] + -1.0f * reward[i] * d;
    d_reward[i] = -d * scale;
    const float dn = done[i] ? 1.0f : 0.0f;
    disc[i] = d * gamma * (1.0f - dn) + dn;
}

// k_bptt_accumulate + the state checkpoint of the NEXT step (a plain slab -> tape row copy) in one launch: both sit between
// env step t and env step t + 1 of the BPTT forward pass, one launch boundary (~5 us at 16 384 agents) instead of two
__global__ __launch_bounds__(kBlock) void k_bptt_accumulate_checkpoint(const float* __restrict__ reward, const uint8_t* __restrict__ done,
                                                                       float* __restrict__ disc, float* __restrict__ loss,
                                                                       float* __restrict__ d_reward, float gamma, float scale, int N,
                                                                       const float4* __restrict__ slab, float4* __restrict__ tape,
                                                                       long long n4)
{
    const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x, stride = (long long)gridDim.x * kBlock;
    for (long long j = tid; j < n4; j += stride) tape[j] = slab[j];
    if (tid < N) {
        const int i = (int)tid;
        const float d = disc[i];
        loss[i] = loss[i] + -1.0f * reward[i] * d;
        d_reward[i] = -d * scale;
        const float dn = done[i] ? 1.0f : 0.0f;
        disc[i] = d * gamma * (1.0f - dn) + dn;
    }
}

__global__ __launch_bounds__(1024) void k_fold_stats(const float* __restrict__ part, int nblk, float* __restrict__ stats,
                                                     float* __restrict__ d_log_std_out, float* __restrict__ stats_accum)
{
    // 16 stats x 64 lanes (1024 threads, one wave per statistic): lane-strided sums over the partial rows, then a
    // shuffle tree over the wave -- fixed order, deterministic
    const int k = threadIdx.x >> 6, sl = threadIdx.x & 63;
    float s = 0.0f;
    if (k < 9)
        for (int b = sl; b < nblk; b += 64) s += part[(size_t)b * kStats + k];
    s = wave_sumf(s);
    if (sl == 0) {
        stats[k] = s;
        if (d_log_std_out && k >= 5 && k < 9) d_log_std_out[k - 5] = s;
        if (stats_accum) stats_accum[k] += s;
    }
}

// dst[i, :] = src[perm[i], :] for every field (blockIdx.y); one thread per output element: coalesced stores, the w
// consecutive words of a source row read by consecutive lanes
__global__ __launch_bounds__(kBlock) void k_gather_rows(const vf_gather_fields f, const int64_t* __restrict__ perm, long rows)
{
    // a row's w floats on 2^ceil(log2 w) consecutive lanes: row / column by shift and mask (the flat form `idx / w` was a 64-bit division
    // per element -- 1.1 ms per shuffle of the 8.4 M-row PPO buffer), the permutation entry is read once per row and broadcast by the cache
    const int fi = blockIdx.y;
    const int w = f.width[fi];
    const float* __restrict__ src = f.src[fi];
    float* __restrict__ dst = f.dst[fi];
    const int lg = w > 1 ? 32 - __clz(w - 1) : 0;
    const int c = (int)(threadIdx.x & ((1u << lg) - 1u));
    const long per_block = kBlock >> lg;                      // rows per block and pass (lg <= 8: VF widths are <= 256)
    for (long r = (long)blockIdx.x * per_block + (threadIdx.x >> lg); r < rows; r += (long)gridDim.x * per_block)
        if (c < w) dst[r * w + c] = src[perm[r] * w + c];
}

// reward + gamma * V(terminal obs) on truncated episodes; next episode_start = float(done)   (SB3 collect_rollouts)
__global__ __launch_bounds__(kBlock) void k_rollout_post(const float* __restrict__ reward, const uint8_t* __restrict__ done,
                                                         const uint8_t* __restrict__ ep_flags, const float* __restrict__ tv, float gamma,
                                                         float* __restrict__ reward_out, float* __restrict__ next_start, int N)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const bool d = done[i] != 0;
    const float trunc = (d && (ep_flags[i] & 2)) ? 1.0f : 0.0f;
    reward_out[i] = reward[i] + (gamma * tv[i]) * trunc;
    next_start[i] = d ? 1.0f : 0.0f;
}

// The same bookkeeping with the TimeLimit bootstrap DEFERRED: instead of evaluating V(terminal observation) for all N agents
// at every step (a second policy forward per rollout step), the rows that need it -- done && truncated, a handful per step -- are
// appended to a compact list (flat buffer index t * N + i and the terminal observation rows); after the rollout ONE value
// forward over the collected rows and k_bootstrap_scatter add gamma * V to the listed rewards.  The list order depends on the
// atomic cursor, the result does not: every entry is independent and names a distinct reward.
__global__ __launch_bounds__(kBlock) void k_rollout_post_collect(const float* __restrict__ reward, const uint8_t* __restrict__ done,
                                                                 const uint8_t* __restrict__ ep_flags, float* __restrict__ reward_out,
                                                                 float* __restrict__ next_start, const float* __restrict__ obs0,
                                                                 const float* __restrict__ obs1, int w0, int w1, int* __restrict__ cursor,
                                                                 int capacity, int* __restrict__ idx_list, float* __restrict__ rows0,
                                                                 float* __restrict__ rows1, int flat_base, int N,
                                                                 const float* __restrict__ ep_return, const int* __restrict__ ep_length,
                                                                 float* __restrict__ stat)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const bool d = done[i] != 0;
    reward_out[i] = reward[i];
    next_start[i] = d ? 1.0f : 0.0f;
    if (d && stat) {   // per-agent episode statistics (PPO._dump_logs): every thread owns its agent's four accumulators, summed once per rollout
        float4* a = reinterpret_cast<float4*>(stat) + i;
        float4 v = *a;
        v.x += 1.0f;
        v.y += ep_return[i];
        v.z += (float)ep_length[i];
        v.w += (ep_flags[i] & VF_EP_SUCCESS) ? 1.0f : 0.0f;
        *a = v;
    }
    if (d && (ep_flags[i] & VF_EP_TRUNCATED)) {
        const int slot = atomicAdd(cursor, 1);
        if (slot < capacity) {
            idx_list[slot] = flat_base + i;
            for (int k = 0; k < w0; ++k) rows0[(size_t)slot * w0 + k] = obs0[(size_t)i * w0 + k];
            for (int k = 0; k < w1; ++k) rows1[(size_t)slot * w1 + k] = obs1[(size_t)i * w1 + k];
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_bootstrap_scatter(const int* __restrict__ idx_list, const float* __restrict__ values,
                                                              int count, float gamma, float* __restrict__ rewards_flat)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= count) return;
    const int at = idx_list[j];
    rewards_flat[at] = rewards_flat[at] + gamma * values[j];      // same rounding as reward + (gamma * tv) of k_rollout_post
}

// episode statistics of one env step for the training log (PPO._dump_logs: rollout/ep_rew_mean, ep_len_mean, success rate;
// PPO.py:392-414): acc += {episodes finished, sum of their returns, sum of their lengths, successes}.  One block, fixed
// reduction order, no host synchronisation in the rollout loop.
__global__ __launch_bounds__(1024) void k_episode_stats(const uint8_t* __restrict__ done, const float* __restrict__ ep_return,
                                                        const int32_t* __restrict__ ep_length, const uint8_t* __restrict__ ep_flags,
                                                        double* __restrict__ acc, int N)
{
    __shared__ double sh[4][16];
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    // branch-free and unrolled: the loads of eight strides are in flight together (a `if (done[i])` around the other
    // three loads made every iteration a dependent round trip: 16 us for 32 768 agents)
    for (int i0 = threadIdx.x; i0 < N; i0 += 8 * 1024) {
        uint8_t d[8], f[8];
        float r[8];
        int32_t l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 1024, ic = i < N ? i : N - 1;
            d[u] = i < N ? done[ic] : (uint8_t)0;
            r[u] = ep_return[ic];
            l[u] = ep_length[ic];
            f[u] = ep_flags[ic];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double on = d[u] ? 1.0 : 0.0;
            v[0] += on;
            v[1] += d[u] ? (double)r[u] : 0.0;
            v[2] += d[u] ? (double)l[u] : 0.0;
            v[3] += (d[u] && (f[u] & VF_EP_SUCCESS)) ? 1.0 : 0.0;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double s = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += sh[threadIdx.x][w];
        acc[threadIdx.x] += t;
    }
}

// sum of squares of a short vector (the flat gradient: tens of thousands of floats) in ONE block: fp64 lane sums,
// fixed-order tree -- one launch instead of the partial/final pair
__global__ __launch_bounds__(1024) void k_sumsq_block(const float* __restrict__ x, long n, float* __restrict__ out)
{
    __shared__ double sh[16];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;       // four independent chains, 16-byte loads (all in flight at once)
    const long n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? n >> 2 : 0;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (long i = threadIdx.x; i < n4; i += 1024) {
        const float4 v = x4[i];
        s0 += (double)v.x * v.x; s1 += (double)v.y * v.y; s2 += (double)v.z * v.z; s3 += (double)v.w * v.w;
    }
    for (long i = 4 * n4 + threadIdx.x; i < n; i += 1024) s0 += (double)x[i] * x[i];
    double ss = wave_sum((s0 + s1) + (s2 + s3));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += sh[w];
        *out = (float)t;
    }
}

// clip_grad_norm_ + Adam with L2 weight decay (torch.optim.Adam semantics), PPO.py:285-292
__global__ __launch_bounds__(kBlock) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, long n, const float* __restrict__ sumsq,
                                                 const vf_adam_cfg c, float bc1, float bc2_sqrt)
{
    // the first element's operands are requested ahead of the norm's reduction: they do not depend on it, and a launch this small is
    // nothing but dependent round trips (r06: 5.1 -> see profiles/r06_fused_tail.txt)
    const long i0 = (long)blockIdx.x * kBlock + threadIdx.x;
    float p0 = 0.0f, g0 = 0.0f, m0 = 0.0f, v0 = 0.0f;
    int4 o0 = make_int4(-1, -1, -1, -1);
    if (i0 < n) {
        p0 = p[i0];
        g0 = g[i0];
        m0 = m[i0];
        v0 = v[i0];
        if (c.pack_map) o0 = reinterpret_cast<const int4*>(c.pack_map)[i0];
    }
    float coef = 1.0f;
    if (c.max_grad_norm > 0.0f) {
        float ss;
        if (c.sumsq_partials) {     // every block sums the fold's partials (+ the uncovered tail) in the same fixed order
            __shared__ double sh[4];
            double a = 0.0;
            for (int i = threadIdx.x; i < c.n_sumsq_partials; i += kBlock) a += c.sumsq_partials[i];
            for (long i = c.sumsq_tail_from + threadIdx.x; i < n; i += kBlock) a += (double)g[i] * (double)g[i];
            a = wave_sum(a);
            if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
            __syncthreads();
            ss = (float)((sh[0] + sh[1]) + (sh[2] + sh[3]));
        } else {
            ss = *sumsq;
        }
        coef = adam_clip_coef(ss, c.max_grad_norm);
    }
    const float step = c.lr / bc1;
    if (i0 < n) {
        const float pn = adam_param(p0, g0, m0, v0, coef, c, step, bc2_sqrt);       // vf_adam_device.hpp
        m[i0] = m0;
        v[i0] = v0;
        p[i0] = pn;
        if (o0.x >= 0) c.packed[o0.x] = pn;
        if (o0.y >= 0) c.packed[o0.y] = pn;
        if (o0.z >= 0) c.packed[o0.z] = pn;
        if (o0.w >= 0) c.packed[o0.w] = pn;
    }
    for (long i = i0 + (long)gridDim.x * kBlock; i < n; i += (long)gridDim.x * kBlock) {
        float mi = m[i], vi = v[i];
        const float pn = adam_param(p[i], g[i], mi, vi, coef, c, step, bc2_sqrt);
        m[i] = mi;
        v[i] = vi;
        p[i] = pn;
        if (c.pack_map) adam_refresh_packed(c, i, pn);
    }
}

inline int grid_for(long n, int cap = 512)
{
    long b = (n + kBlock - 1) / kBlock;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace vf

namespace {

size_t linear_lds_bytes(bool bwd, int K, int No)
{
    const int red = bwd ? No : K, cols = bwd ? K : No;
    const int red16 = (red + 15) & ~15, ct = (cols + 31) >> 5, sa = red16 + 1;
    const size_t ws = bwd ? (size_t)red16 * (ct * 32 + 1) : (size_t)ct * 32 * sa;
    return ((size_t)vf::kRows * sa + ws) * sizeof(float);
}

template <typename Kern>
int allow_lds(Kern k, size_t bytes)
{
    if (bytes > 64 * 1024) VF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return VF_OK;
}

int linear_grid(int M)
{
    const int ntiles = (M + vf::kRows - 1) / vf::kRows;
    return ntiles < 512 ? ntiles : 512;   // weight-stationary blocks, two per CU
}

int wgrad_rows_per_block(int M)
{
    int rpb = (M + 255) / 256;  // aim at <= 256 chunks (one per CU), each a multiple of the 64-row tile
    rpb = (rpb + vf::kRows - 1) / vf::kRows * vf::kRows;
    return rpb < vf::kRows ? vf::kRows : rpb;
}

}  // namespace

extern "C" {

int vf_gae(const float* rewards, const float* values, const float* episode_starts, const float* last_values,
           const float* dones, float* adv, float* ret, int32_t T, int32_t N, double gamma, double lam, vf_stream_t stream)
{
    if (!rewards || !values || !episode_starts || !last_values || !dones || !adv || !ret || T <= 0 || N <= 0)
        return vf::fail(VF_EINVAL, "vf_gae: bad argument");
    hipLaunchKernelGGL(vf::k_gae, dim3(vf::blocks_for(N)), dim3(vf::kBlock), 0, vf::as_stream(stream), rewards, values,
                       episode_starts, last_values, dones, adv, ret, T, N, (float)gamma, (float)(gamma * lam));
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_td_returns(const float* r, const uint8_t* done, const uint8_t* episode_done, const float* next_value, float* returns,
                  int32_t H, int32_t N, double gamma, double lamda, vf_stream_t stream)
{
    if (!r || !done || !next_value || !returns || H <= 0 || N <= 0) return vf::fail(VF_EINVAL, "vf_td_returns: bad argument");
    hipLaunchKernelGGL(vf::k_td_returns, dim3(vf::blocks_for(N)), dim3(vf::kBlock), 0, vf::as_stream(stream), r, done,
                       episode_done ? episode_done : done, next_value, returns, H, N, (float)gamma, (float)lamda,
                       (float)(lamda * gamma), (float)(1.0 - lamda));
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_adv_normalize(const float* adv, float* out, int64_t n, int64_t count, double* sums_inout, float* scratch,
                     int32_t phase, vf_stream_t stream)
{
    if (!adv || !out || !scratch || n <= 0 || count < 2 || phase < 0 || phase > 2)
        return vf::fail(VF_EINVAL, "vf_adv_normalize: bad argument");
    hipStream_t st = vf::as_stream(stream);
    double* part = reinterpret_cast<double*>(scratch);  // [2*nblk] + 2 doubles for the sums
    const int nblk = vf::grid_for(n, 256);
    double* sums = sums_inout ? sums_inout : part + 2 * 256;
    if (phase == 0 || phase == 2) {
        hipLaunchKernelGGL(vf::k_sum2_partial, dim3(nblk), dim3(vf::kBlock), 0, st, adv, (long)n, part);
        hipLaunchKernelGGL(vf::k_sum2_final, dim3(1), dim3(64), 0, st, part, nblk, sums, (float*)nullptr);
    }
    if (phase == 1 || phase == 2)
        hipLaunchKernelGGL(vf::k_adv_apply, dim3(vf::grid_for(n, 1024)), dim3(vf::kBlock), 0, st, adv, out, (long)n, sums,
                           (double)count);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_adv_normalize_segments(const float* adv, float* out, int32_t n_seg, int64_t seg_len, int64_t count, double* sums,
                              int32_t phase, vf_stream_t stream)
{
    if (!adv || !out || !sums || n_seg <= 0 || n_seg > 65535 || seg_len <= 0 || count < 2 || phase < 0 || phase > 2)
        return vf::fail(VF_EINVAL, "vf_adv_normalize_segments: bad argument");
    hipStream_t st = vf::as_stream(stream);
    if (phase == 0 || phase == 2)
        hipLaunchKernelGGL(vf::k_adv_seg_sums, dim3(n_seg), dim3(vf::kBlock), 0, st, adv, (long)seg_len, sums);
    if (phase == 1 || phase == 2) {
        const int bx = (int)((seg_len + 4 * vf::kBlock - 1) / (4 * vf::kBlock));
        hipLaunchKernelGGL(vf::k_adv_seg_apply, dim3(bx < 64 ? bx : 64, n_seg), dim3(vf::kBlock), 0, st, adv, out, (long)seg_len,
                           sums, (double)count);
    }
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_linear_fwd(const float* X, int32_t ldx, const float* W, const float* b, float* Y, int32_t ldy, int32_t M,
                  int32_t K, int32_t No, int32_t relu, vf_stream_t stream)
{
    if (!X || !W || !Y || M <= 0 || K <= 0 || No <= 0 || K > 128 || No > 128 || ldx < K || ldy < No)
        return vf::fail(VF_EINVAL, "vf_linear_fwd: bad argument (K, No <= 128)");
    const size_t lds = linear_lds_bytes(false, K, No);
    const dim3 grid(linear_grid(M)), block(vf::kBlock);
    hipStream_t st = vf::as_stream(stream);
    if (relu < 0 || relu > VF_ACTIVATION_LEAKY_RELU) return vf::fail(VF_EINVAL, "vf_linear_fwd: activation kind %d", relu);
    if (int rc = allow_lds(vf::k_linear<false>, lds)) return rc;
    hipLaunchKernelGGL((vf::k_linear<false>), grid, block, lds, st, X, ldx, (const float*)nullptr, 0, W, b, Y, ldy, M, K, No, 0, relu);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_linear_bwd_data(const float* dY, int32_t lddy, const float* Ymask, int32_t ldym, const float* W, float* dX,
                       int32_t lddx, int32_t M, int32_t K, int32_t No, int32_t accumulate, int32_t act, vf_stream_t stream)
{
    if (!dY || !W || !dX || M <= 0 || K <= 0 || No <= 0 || K > 128 || No > 128 || lddy < No || lddx < K || act < 0 ||
        act > VF_ACTIVATION_LEAKY_RELU)
        return vf::fail(VF_EINVAL, "vf_linear_bwd_data: bad argument (K, No <= 128)");
    const size_t lds = linear_lds_bytes(true, K, No);
    if (int rc = allow_lds(vf::k_linear<true>, lds)) return rc;
    hipLaunchKernelGGL((vf::k_linear<true>), dim3(linear_grid(M)), dim3(vf::kBlock), lds,
                       vf::as_stream(stream), dY, lddy, Ymask, ldym, W, (const float*)nullptr, dX, lddx, M, K, No, accumulate, act);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int64_t vf_linear_bwd_scratch_floats(int32_t M, int32_t K, int32_t No)
{
    const int rpb = wgrad_rows_per_block(M);
    const int nblk = (M + rpb - 1) / rpb;
    return (int64_t)nblk * ((int64_t)No * K + No);
}

static int linear_bwd_weight(const float* dY, int32_t lddy, const float* Ymask, int32_t ldym, const float* X, int32_t ldx,
                             float* dW, float* db, int32_t M, int32_t K, int32_t No, float* scratch, vf_stream_t stream,
                             int accumulate, int act)
{
    if (!dY || !X || !dW || !scratch || M <= 0 || K <= 0 || No <= 0 || K > 128 || No > 128 || act < 0 || act > VF_ACTIVATION_LEAKY_RELU)
        return vf::fail(VF_EINVAL, "vf_linear_bwd_weight: bad argument (K, No <= 128)");
    const int rpb = wgrad_rows_per_block(M);
    const int nblk = (M + rpb - 1) / rpb;
    const int nt = (No + 31) >> 5, kt = (K + 31) >> 5;
    const size_t lds = (size_t)vf::kRows * ((nt * 32 + 1) + (kt * 32 + 1)) * sizeof(float);
    if (int rc = allow_lds(vf::k_linear_wgrad, lds)) return rc;
    hipStream_t st = vf::as_stream(stream);
    hipLaunchKernelGGL(vf::k_linear_wgrad, dim3(nblk), dim3(vf::kBlock), lds, st, dY, lddy, Ymask, ldym, X, ldx, scratch, M, K,
                       No, rpb, act);
    const int n = No * K + No;
    hipLaunchKernelGGL(vf::k_fold_partials, dim3((n + 63) / 64), dim3(vf::kBlock), 0, st, scratch, nblk, n,
                       No * K, No, dW, db, accumulate);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_linear_bwd_weight(const float* dY, int32_t lddy, const float* Ymask, int32_t ldym, const float* X, int32_t ldx,
                         float* dW, float* db, int32_t M, int32_t K, int32_t No, float* scratch, int32_t act, vf_stream_t stream)
{
    return linear_bwd_weight(dY, lddy, Ymask, ldym, X, ldx, dW, db, M, K, No, scratch, stream, 0, act);
}

int vf_linear_bwd_weight_acc(const float* dY, int32_t lddy, const float* Ymask, int32_t ldym, const float* X, int32_t ldx,
                             float* dW, float* db, int32_t M, int32_t K, int32_t No, float* scratch, int32_t act, vf_stream_t stream)
{
    return linear_bwd_weight(dY, lddy, Ymask, ldym, X, ldx, dW, db, M, K, No, scratch, stream, 1, act);
}

int64_t vf_mlp_packed_floats(const vf_mlp_desc* desc)
{
    if (!desc || desc->n_layers < 1 || desc->n_layers > VF_MLP_MAX_LAYERS) return -1;
    int64_t n = 0;
    for (int i = 0; i < desc->n_layers; ++i) {
        const vf_mlp_layer& L = desc->layer[i];
        const int64_t end = L.wt_off + (int64_t)((L.K + 15) & ~15) * ((L.No + 31) & ~31);
        const int64_t endb = L.wb_off + (int64_t)((L.No + 15) & ~15) * ((L.K + 31) & ~31);
        const int64_t G = L.src < 4 ? (L.K + 7) >> 3 : ((L.K + 31) >> 5) * 4;
        const int64_t endr = L.wr_off + (int64_t)((L.No + 31) >> 5) * G * 256;
        n = end > n ? end : n;
        n = endb > n ? endb : n;
        const int64_t endq = L.wq_off + (int64_t)((L.K + 31) >> 5) * ((L.No + 7) >> 3) * 256;
        n = endr > n ? endr : n;
        n = endq > n ? endq : n;
    }
    return n;
}

int vf_mlp_pack_weights(const vf_mlp_desc* desc, const float* params, float* packed, vf_stream_t stream)
{
    if (!desc || !params || !packed || desc->n_layers < 1 || desc->n_layers > VF_MLP_MAX_LAYERS)
        return vf::fail(VF_EINVAL, "vf_mlp_pack_weights: bad argument");
    hipLaunchKernelGGL(vf::k_mlp_pack_weights, dim3(8, desc->n_layers), dim3(vf::kBlock), 0, vf::as_stream(stream), *desc, params,
                       packed);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_mlp_forward(const vf_mlp_desc* desc, const float* params, const float* packed, const float* in0, const float* in1,
                   const float* in2, const float* in3, float* out0, float* out1, int32_t M, vf_stream_t stream)
{
    if (!desc || !params || !packed || !in0 || M <= 0) return vf::fail(VF_EINVAL, "vf_mlp_forward: bad argument");
    if (desc->n_layers < 1 || desc->n_layers > VF_MLP_MAX_LAYERS || desc->n_inputs < 1 || desc->n_inputs > 4)
        return vf::fail(VF_EINVAL, "vf_mlp_forward: bad layer / input count");
    for (int i = 0; i < desc->n_layers; ++i) {
        const vf_mlp_layer& L = desc->layer[i];
        if (L.K < 1 || L.K > 128 || L.No < 1 || L.No > 128) return vf::fail(VF_EINVAL, "vf_mlp_forward: layer %d: K, No must be 1..128", i);
    }
    // reference-default network shapes: activations chained through MFMA accumulator registers (vf_mlp_chain.hip);
    // out1 == NULL there means "skip the value trunk"
    if (int rc = vf::mlp_forward_chain_try(desc, params, packed, in0, in1, out0, out1, M, vf::as_stream(stream), nullptr, in2)) return rc < 0 ? rc : VF_OK;
    for (int i = 0; i < desc->n_layers; ++i) {
        const vf_mlp_layer& L = desc->layer[i];
        if (L.dst >= VF_MLP_OUT0 && !(L.dst == VF_MLP_OUT0 ? out0 : out1))
            return vf::fail(L.dst == VF_MLP_OUT1 ? VF_EUNSUPPORTED : VF_EINVAL, "vf_mlp_forward: missing output %d", L.dst);
    }
    const size_t lds = (size_t)desc->lds_floats * sizeof(float);
    if (lds > 160 * 1024) return vf::fail(VF_EINVAL, "vf_mlp_forward: LDS plan needs %zu bytes (> 160 KiB)", lds);
    if (int rc = allow_lds(vf::k_mlp_forward, lds)) return rc;
    const int ntiles = (M + vf::kRows - 1) / vf::kRows;
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;           // workgroups that fit one CU's 160 KiB
    const int cap = 256 * per_cu;
    vf::MlpIo io{{in0, in1, in2, in3}, {out0, out1}};
    hipLaunchKernelGGL(vf::k_mlp_forward, dim3(ntiles < cap ? ntiles : cap), dim3(vf::kBlock), lds, vf::as_stream(stream), *desc,
                       params, packed, io, M);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

#ifdef VF_PROBE
int vf_probe_read(unsigned long long* out, int32_t reset)
{
    VF_HIP(hipDeviceSynchronize());
    VF_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(vf::g_probe), sizeof(unsigned long long) * 32));
    if (reset) {
        unsigned long long z[32] = {0};
        VF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(vf::g_probe), z, sizeof(z)));
    }
    return VF_OK;
}
#endif

int32_t vf_mlp_backward_blocks(int32_t M)
{
    if (M <= 0) return 0;
    const int mtiles = (M + vf::kRows - 1) / vf::kRows;
    const int rounds = (mtiles + 255) / 256;      // tiles per block: equal work, at most one (8-wave) block per CU
    return (mtiles + rounds - 1) / rounds;
}

int64_t vf_mlp_backward_partial_floats(const vf_mlp_bwd_desc* desc, int32_t M)
{
    if (!desc || desc->n_layers < 1 || desc->n_layers > VF_MLP_MAX_LAYERS || M <= 0) return -1;
    const int64_t a = (int64_t)vf_mlp_backward_blocks(M) * desc->n_fold, b = vf::mlp_wgrad_partial_floats(desc, M);
    return a > b ? a : b;
}

int vf_mlp_backward(const vf_mlp_bwd_desc* desc, const float* packed, float* partials, float* grad, int32_t M,
                    int32_t accumulate, vf_stream_t stream)
{
    if (!desc || !packed || !partials || !grad || M <= 0) return vf::fail(VF_EINVAL, "vf_mlp_backward: bad argument");
    if (desc->n_layers < 1 || desc->n_layers > VF_MLP_MAX_LAYERS || desc->n_fold < 1)
        return vf::fail(VF_EINVAL, "vf_mlp_backward: bad layer count / n_fold");
    size_t lds = 0;
    for (int i = 0; i < desc->n_layers; ++i) {
        const vf_mlp_bwd_layer& L = desc->layer[i];
        if (L.K < 1 || L.K > 128 || L.No < 1 || L.No > 128) return vf::fail(VF_EINVAL, "vf_mlp_backward: layer %d: K, No must be 1..128", i);
        if (!L.dY || !L.X || L.ld_dy < L.No || L.ld_x < L.K || (L.Y && L.ld_y < L.No) || (L.need_dx && (!L.dX || L.ld_dx < L.K)))
            return vf::fail(VF_EINVAL, "vf_mlp_backward: layer %d: missing pointer or short row stride", i);
        if (L.w_off < 0 || L.b_off < 0 || L.w_off + (int64_t)L.K * L.No > desc->n_fold || L.b_off + L.No > desc->n_fold)
            return vf::fail(VF_EINVAL, "vf_mlp_backward: layer %d: parameter offsets outside n_fold", i);
    }
    // reference-default network classes: reverse chain in registers + row-slab weight gradients (vf_mlp_chain.hip, vf_mlp_wgrad.hip)
    if (int rc = vf::mlp_backward_chain_try(desc, packed, M, vf::as_stream(stream))) {
        if (rc < 0) return rc;
        return vf::mlp_wgrad_launch(desc, partials, grad, M, accumulate, nullptr, nullptr, vf::as_stream(stream));
    }
    lds = ((size_t)2 * vf::kRows * (129 + 129) + vf::kBwdThreads) * sizeof(float);   // two staging buffers + bias scratch
    if (int rc = allow_lds(vf::k_mlp_backward, lds)) return rc;
    const int nblk = vf_mlp_backward_blocks(M);
    hipStream_t st = vf::as_stream(stream);
    hipLaunchKernelGGL(vf::k_mlp_backward, dim3(nblk), dim3(vf::kBwdThreads), lds, st, *desc, packed, partials, M);
    // fold the parameter ranges the listed layers cover (a skipped trunk leaves its columns of `partials` unwritten)
    std::pair<int64_t, int64_t> iv[2 * VF_MLP_MAX_LAYERS];
    int niv = 0;
    for (int i = 0; i < desc->n_layers; ++i) {
        const vf_mlp_bwd_layer& L = desc->layer[i];
        iv[niv++] = {L.w_off, L.w_off + (int64_t)L.K * L.No};
        iv[niv++] = {L.b_off, L.b_off + L.No};
    }
    std::sort(iv, iv + niv);
    for (int i = 0; i < niv;) {
        int64_t lo = iv[i].first, hi = iv[i].second;
        int j = i + 1;
        while (j < niv && iv[j].first <= hi) { hi = iv[j].second > hi ? iv[j].second : hi; ++j; }
        const int n = (int)(hi - lo);
        hipLaunchKernelGGL(vf::k_fold_partials, dim3((n + 63) / 64), dim3(vf::kBlock), 0, st, partials + lo, nblk, desc->n_fold, n, 0,
                           grad + lo, (float*)nullptr, accumulate ? 1 : 0);
        i = j;
    }
    VF_HIP(hipGetLastError());
    return VF_OK;
}

static int check_bwd_desc(const vf_mlp_bwd_desc* desc, const char* who)
{
    if (!desc || desc->n_layers < 1 || desc->n_layers > VF_MLP_MAX_LAYERS || desc->n_fold < 1) return vf::fail(VF_EINVAL, "%s: bad layer count / n_fold", who);
    for (int i = 0; i < desc->n_layers; ++i) {
        const vf_mlp_bwd_layer& L = desc->layer[i];
        if (L.K < 1 || L.K > 128 || L.No < 1 || L.No > 128) return vf::fail(VF_EINVAL, "%s: layer %d: K, No must be 1..128", who, i);
        if (!L.dY || !L.X || L.ld_dy < L.No || L.ld_x < L.K || (L.Y && L.ld_y < L.No) || (L.need_dx && (!L.dX || L.ld_dx < L.K)))
            return vf::fail(VF_EINVAL, "%s: layer %d: missing pointer or short row stride", who, i);
        if (L.w_off < 0 || L.b_off < 0 || L.w_off + (int64_t)L.K * L.No > desc->n_fold || L.b_off + L.No > desc->n_fold)
            return vf::fail(VF_EINVAL, "%s: layer %d: parameter offsets outside n_fold", who, i);
    }
    return VF_OK;
}

int vf_mlp_backward_data_supported(const vf_mlp_bwd_desc* desc)
{
    if (check_bwd_desc(desc, "vf_mlp_backward_data_supported")) return 0;
    return vf::mlp_backward_chain_try(desc, nullptr, 1, nullptr) == 1 ? 1 : 0;
}

int vf_mlp_backward_data(const vf_mlp_bwd_desc* desc, const float* packed, int32_t M, vf_stream_t stream)
{
    if (!packed || M <= 0) return vf::fail(VF_EINVAL, "vf_mlp_backward_data: bad argument");
    if (int rc = check_bwd_desc(desc, "vf_mlp_backward_data")) return rc;
    const int rc = vf::mlp_backward_chain_try(desc, packed, M, vf::as_stream(stream));
    if (rc < 0) return rc;
    if (rc == 0) return vf::fail(VF_EUNSUPPORTED, "vf_mlp_backward_data: the layer table is not an instantiated network class / variant");
    return VF_OK;
}

int vf_mlp_forward_steps(const vf_mlp_desc* desc, const float* params, const float* packed, const float* in0, const float* in1,
                         const float* in2, float* out0, float* out1, int32_t M_step, int32_t n_steps, vf_stream_t stream)
{
    if (!desc || !params || !packed || !in0 || !out0 || M_step <= 0 || n_steps <= 0) return vf::fail(VF_EINVAL, "vf_mlp_forward_steps: bad argument");
    if (desc->n_layers < 1 || desc->n_layers > VF_MLP_MAX_LAYERS || desc->n_inputs < 1 || desc->n_inputs > 4)
        return vf::fail(VF_EINVAL, "vf_mlp_forward_steps: bad layer / input count");
    if (M_step & 31) return vf::fail(VF_EUNSUPPORTED, "vf_mlp_forward_steps: M_step must be a multiple of 32 (whole row tiles per step)");
    if ((int64_t)M_step * n_steps > 0x7fffffff) return vf::fail(VF_EINVAL, "vf_mlp_forward_steps: M_step x n_steps passes 2^31 rows");
    for (int i = 0; i < desc->n_layers; ++i)
        if (desc->layer[i].save) return vf::fail(VF_EINVAL, "vf_mlp_forward_steps: inference only (no saved activations)");
    const int rc = vf::mlp_forward_chain_try(desc, params, packed, in0, in1, out0, out1, M_step * n_steps, vf::as_stream(stream), nullptr, in2, M_step);
    if (rc < 0) return rc;
    if (rc == 0) return vf::fail(VF_EUNSUPPORTED, "vf_mlp_forward_steps: the layer table is not an instantiated network class");
    return VF_OK;
}

int vf_mlp_forward_act(const vf_mlp_desc* desc, const float* params, const float* packed, const float* in0, const float* in1,
                       const float* log_std, const float* eps, float* action, float* obs_copy0, float* obs_copy1, int32_t M,
                       vf_stream_t stream)
{
    if (!desc || !params || !packed || !in0 || !log_std || !eps || !action || M <= 0 || desc->n_layers < 1 || desc->n_layers > VF_MLP_MAX_LAYERS)
        return vf::fail(VF_EINVAL, "vf_mlp_forward_act: bad argument");
    const vf::ReparamFwd rp{log_std, eps, action, {obs_copy0, obs_copy1}};
    const int rc = vf::mlp_forward_chain_try(desc, params, packed, in0, in1, nullptr, nullptr, M, vf::as_stream(stream), &rp);
    if (rc < 0) return rc;
    if (rc == 0) return vf::fail(VF_EUNSUPPORTED, "vf_mlp_forward_act: the layer table is not an instantiated network class");
    return VF_OK;
}

int vf_mlp_backward_data_act(const vf_mlp_bwd_desc* desc, const float* packed, const float* d_action, const float* action,
                             const float* log_std, const float* eps, float* g_log_std, int32_t M, vf_stream_t stream)
{
    if (!packed || !d_action || !action || !log_std || !eps || !g_log_std || M <= 0) return vf::fail(VF_EINVAL, "vf_mlp_backward_data_act: bad argument");
    if (int rc = check_bwd_desc(desc, "vf_mlp_backward_data_act")) return rc;
    const vf::ReparamBwd rp{d_action, action, log_std, eps, g_log_std};
    const int rc = vf::mlp_backward_chain_try(desc, packed, M, vf::as_stream(stream), &rp);
    if (rc < 0) return rc;
    if (rc == 0) return vf::fail(VF_EUNSUPPORTED, "vf_mlp_backward_data_act: the layer table is not an instantiated network class / variant");
    return VF_OK;
}

int vf_mlp_weight_grad(const vf_mlp_bwd_desc* desc, float* partials, float* grad, int32_t M, int32_t accumulate, vf_stream_t stream)
{
    if (!partials || !grad || M <= 0) return vf::fail(VF_EINVAL, "vf_mlp_weight_grad: bad argument");
    if (int rc = check_bwd_desc(desc, "vf_mlp_weight_grad")) return rc;
    return vf::mlp_wgrad_launch(desc, partials, grad, M, accumulate, nullptr, nullptr, vf::as_stream(stream));
}

int32_t vf_mlp_weight_grad_fold_blocks(const vf_mlp_bwd_desc* desc)
{
    if (check_bwd_desc(desc, "vf_mlp_weight_grad_fold_blocks")) return -1;
    return vf::mlp_wgrad_fold_blocks(desc);
}

int vf_mlp_weight_grad_sumsq(const vf_mlp_bwd_desc* desc, float* partials, float* grad, int32_t M, int32_t accumulate,
                             double* sumsq_partials, const vf_stats_fold* loss_stats, vf_stream_t stream)
{
    if (!partials || !grad || !sumsq_partials || M <= 0) return vf::fail(VF_EINVAL, "vf_mlp_weight_grad_sumsq: bad argument");
    if (loss_stats && (!loss_stats->part || !loss_stats->stats || loss_stats->n_rows < 1))
        return vf::fail(VF_EINVAL, "vf_mlp_weight_grad_sumsq: bad loss_stats");
    if (int rc = check_bwd_desc(desc, "vf_mlp_weight_grad_sumsq")) return rc;
    return vf::mlp_wgrad_launch(desc, partials, grad, M, accumulate, sumsq_partials, loss_stats, vf::as_stream(stream));
}

int vf_mlp_weight_grad_layers(const vf_mlp_bwd_desc* desc, float* partials, float* grad, int32_t M, int32_t accumulate, uint32_t layer_mask,
                              vf_stream_t stream)
{
    if (!partials || !grad || M <= 0) return vf::fail(VF_EINVAL, "vf_mlp_weight_grad_layers: bad argument");
    if (int rc = check_bwd_desc(desc, "vf_mlp_weight_grad_layers")) return rc;
    return vf::mlp_wgrad_launch_layers(desc, partials, grad, M, accumulate, layer_mask, vf::as_stream(stream));
}

int vf_mlp_weight_grad_adam(const vf_mlp_bwd_desc* desc, float* partials, float* grad, int32_t M, int32_t accumulate,
                            const vf_stats_fold* loss_stats, const vf_wgrad_tail* tail, vf_stream_t stream)
{
    if (!partials || !grad || !tail || M <= 0) return vf::fail(VF_EINVAL, "vf_mlp_weight_grad_adam: bad argument");
    if (!tail->param || !tail->exp_avg || !tail->exp_avg_sq || !tail->sync || !tail->adam.sumsq_partials || tail->n <= 0 ||
        tail->adam.step <= 0 || tail->adam.sumsq_tail_from < 0 || tail->adam.sumsq_tail_from > tail->n ||
        (reinterpret_cast<uintptr_t>(tail->sync) & 63) || (reinterpret_cast<uintptr_t>(tail->adam.sumsq_partials) & 7))
        return vf::fail(VF_EINVAL, "vf_mlp_weight_grad_adam: bad tail (sync: 64-byte aligned)");
    if ((tail->adam.pack_map == nullptr) != (tail->adam.packed == nullptr))
        return vf::fail(VF_EINVAL, "vf_mlp_weight_grad_adam: pack_map and packed must be given together");
    if (loss_stats && (!loss_stats->part || !loss_stats->stats || loss_stats->n_rows < 1 || (reinterpret_cast<uintptr_t>(loss_stats->part) & 15)))
        return vf::fail(VF_EINVAL, "vf_mlp_weight_grad_adam: bad loss_stats (part: 16-byte aligned rows)");
    if (int rc = check_bwd_desc(desc, "vf_mlp_weight_grad_adam")) return rc;
    const int rc = vf::mlp_wgrad_adam_launch(desc, partials, grad, M, accumulate, loss_stats, tail, vf::as_stream(stream));
    if (rc < 0) return rc;
    return rc == 1 ? VF_OK : VF_EUNSUPPORTED;      // the reason is in vf_last_error()
}

int vf_reparam_fwd(const float* mean, const float* log_std, const float* eps, float* action, int32_t N, vf_stream_t stream)
{
    if (!mean || !log_std || !eps || !action || N <= 0) return vf::fail(VF_EINVAL, "vf_reparam_fwd: bad argument");
    hipLaunchKernelGGL(vf::k_reparam_fwd, dim3(vf::blocks_for(N)), dim3(vf::kBlock), 0, vf::as_stream(stream),
                       reinterpret_cast<const float4*>(mean), log_std, reinterpret_cast<const float4*>(eps),
                       reinterpret_cast<float4*>(action), N);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_reparam_bwd(const float* d_action, const float* action, const float* log_std, const float* eps, float* d_mean,
                   float* g_log_std, int32_t N, vf_stream_t stream)
{
    if (!d_action || !action || !log_std || !eps || !d_mean || !g_log_std || N <= 0)
        return vf::fail(VF_EINVAL, "vf_reparam_bwd: bad argument");
    hipLaunchKernelGGL(vf::k_reparam_bwd, dim3(vf::blocks_for(N)), dim3(vf::kBlock), 0, vf::as_stream(stream),
                       reinterpret_cast<const float4*>(d_action), reinterpret_cast<const float4*>(action), log_std,
                       reinterpret_cast<const float4*>(eps), reinterpret_cast<float4*>(d_mean),
                       reinterpret_cast<float4*>(g_log_std), N);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_bptt_accumulate(const float* reward, const uint8_t* done, float* disc, float* loss, float* d_reward, float gamma,
                       float scale, int32_t N, vf_stream_t stream)
{
    if (!reward || !done || !disc || !loss || !d_reward || N <= 0) return vf::fail(VF_EINVAL, "vf_bptt_accumulate: bad argument");
    hipLaunchKernelGGL(vf::k_bptt_accumulate, dim3(vf::blocks_for(N)), dim3(vf::kBlock), 0, vf::as_stream(stream), reward, done,
                       disc, loss, d_reward, gamma, scale, N);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_bptt_accumulate_checkpoint(const float* reward, const uint8_t* done, float* disc, float* loss, float* d_reward, float gamma,
                                  float scale, int32_t N, const float* slab, float* tape_row, int64_t slab_floats, vf_stream_t stream)
{
    if (!reward || !done || !disc || !loss || !d_reward || N <= 0 || !slab || !tape_row || slab_floats <= 0 || (slab_floats & 3))
        return vf::fail(VF_EINVAL, "vf_bptt_accumulate_checkpoint: bad argument");
    if ((reinterpret_cast<uintptr_t>(slab) | reinterpret_cast<uintptr_t>(tape_row)) & 15)
        return vf::fail(VF_EINVAL, "vf_bptt_accumulate_checkpoint: slab and tape row must be 16-byte aligned");
    const long long n4 = slab_floats / 4;
    long long blocks = (n4 + vf::kBlock - 1) / vf::kBlock;
    if (blocks < vf::blocks_for(N)) blocks = vf::blocks_for(N);
    if (blocks > 4096) blocks = 4096;                    // grid-stride copy; 4096 x 256 threads cover the accumulate for N <= 1 M
    if ((long long)N > blocks * vf::kBlock) return vf::fail(VF_EINVAL, "vf_bptt_accumulate_checkpoint: N too large for one launch");
    hipLaunchKernelGGL(vf::k_bptt_accumulate_checkpoint, dim3((unsigned)blocks), dim3(vf::kBlock), 0, vf::as_stream(stream), reward, done,
                       disc, loss, d_reward, gamma, scale, N, reinterpret_cast<const float4*>(slab), reinterpret_cast<float4*>(tape_row), n4);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_episode_stats(const uint8_t* done, const float* ep_return, const int32_t* ep_length, const uint8_t* ep_flags, double* acc4,
                     int32_t N, vf_stream_t stream)
{
    if (!done || !ep_return || !ep_length || !ep_flags || !acc4 || N <= 0) return vf::fail(VF_EINVAL, "vf_episode_stats: bad argument");
    hipLaunchKernelGGL(vf::k_episode_stats, dim3(1), dim3(1024), 0, vf::as_stream(stream), done, ep_return, ep_length, ep_flags, acc4, N);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_head_sample(const float* mean, const float* log_std, float* action, float* log_prob, int32_t M, uint64_t seed,
                   uint64_t step, int32_t deterministic, vf_stream_t stream)
{
    if (!mean || !log_std || !action || !log_prob || M <= 0) return vf::fail(VF_EINVAL, "vf_head_sample: bad argument");
    hipLaunchKernelGGL(vf::k_head_sample, dim3(vf::blocks_for(M)), dim3(vf::kBlock), 0, vf::as_stream(stream),
                       reinterpret_cast<const float4*>(mean), log_std, reinterpret_cast<float4*>(action), log_prob, M,
                       (unsigned long long)seed, (unsigned long long)step, deterministic);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_ppo_loss(const float* mean, const float* value, const float* log_std, const float* action, const float* old_log_prob,
                const float* adv, const float* ret, float* d_mean, float* d_value, float* stats, int32_t M,
                const vf_ppo_loss_cfg* cfg, float* scratch, vf_stream_t stream)
{
    if (!mean || !value || !log_std || !action || !old_log_prob || !adv || !ret || !d_mean || !d_value || !stats || !cfg ||
        !scratch || M <= 0)
        return vf::fail(VF_EINVAL, "vf_ppo_loss: bad argument");
    const int nblk = vf::blocks_for(M);
    if (nblk > 1024) return vf::fail(VF_EINVAL, "vf_ppo_loss: at most 262144 rows per call (scratch contract)");
    hipStream_t st = vf::as_stream(stream);
    hipLaunchKernelGGL(vf::k_ppo_loss, dim3(nblk), dim3(vf::kBlock), 0, st, reinterpret_cast<const float4*>(mean), value,
                       log_std, reinterpret_cast<const float4*>(action), old_log_prob, adv, ret,
                       reinterpret_cast<float4*>(d_mean), d_value, scratch, M, *cfg);
    hipLaunchKernelGGL(vf::k_fold_stats, dim3(1), dim3(1024), 0, st, scratch, nblk, stats, cfg->d_log_std_out, cfg->stats_accum);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_gather_rows(const vf_gather_fields* fields, const int64_t* perm, int64_t rows, vf_stream_t stream)
{
    if (!fields || !perm || rows <= 0 || fields->n_fields < 1 || fields->n_fields > VF_GATHER_MAX_FIELDS)
        return vf::fail(VF_EINVAL, "vf_gather_rows: bad argument");
    int wmax = 1;
    for (int i = 0; i < fields->n_fields; ++i) {
        if (!fields->src[i] || !fields->dst[i] || fields->width[i] < 1) return vf::fail(VF_EINVAL, "vf_gather_rows: field %d: bad pointer / width", i);
        wmax = fields->width[i] > wmax ? fields->width[i] : wmax;
    }
    if (wmax > vf::kBlock) return vf::fail(VF_EINVAL, "vf_gather_rows: rows wider than %d floats", vf::kBlock);
    int wp = 1;
    while (wp < wmax) wp <<= 1;
    const long blocks = (rows * wp + vf::kBlock - 1) / vf::kBlock;
    hipLaunchKernelGGL(vf::k_gather_rows, dim3((unsigned)(blocks < 65536 ? blocks : 65536), fields->n_fields), dim3(vf::kBlock), 0,
                       vf::as_stream(stream), *fields, perm, (long)rows);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_rollout_post(const float* reward, const uint8_t* done, const uint8_t* ep_flags, const float* terminal_value, float gamma,
                    float* reward_out, float* next_episode_start, int32_t N, vf_stream_t stream)
{
    if (!reward || !done || !ep_flags || !terminal_value || !reward_out || !next_episode_start || N <= 0)
        return vf::fail(VF_EINVAL, "vf_rollout_post: bad argument");
    hipLaunchKernelGGL(vf::k_rollout_post, dim3(vf::blocks_for(N)), dim3(vf::kBlock), 0, vf::as_stream(stream), reward, done, ep_flags,
                       terminal_value, gamma, reward_out, next_episode_start, N);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_rollout_post_collect(const float* reward, const uint8_t* done, const uint8_t* ep_flags, float* reward_out,
                            float* next_episode_start, const float* obs0, const float* obs1, int32_t w0, int32_t w1, int32_t* cursor,
                            int32_t capacity, int32_t* idx_list, float* rows0, float* rows1, int32_t flat_base, int32_t N,
                            const float* ep_return, const int32_t* ep_length, float* episode_stat, vf_stream_t stream)
{
    if (!reward || !done || !ep_flags || !reward_out || !next_episode_start || !obs0 || !cursor || !idx_list || !rows0 || N <= 0 ||
        w0 <= 0 || w1 < 0 || capacity <= 0 || (w1 > 0 && (!obs1 || !rows1)) || (episode_stat && (!ep_return || !ep_length)))
        return vf::fail(VF_EINVAL, "vf_rollout_post_collect: bad argument");
    if (episode_stat && reinterpret_cast<uintptr_t>(episode_stat) % 16) return vf::fail(VF_EINVAL, "vf_rollout_post_collect: episode_stat must be 16-byte aligned");
    hipLaunchKernelGGL(vf::k_rollout_post_collect, dim3(vf::blocks_for(N)), dim3(vf::kBlock), 0, vf::as_stream(stream), reward, done,
                       ep_flags, reward_out, next_episode_start, obs0, obs1, w0, w1, cursor, capacity, idx_list, rows0, rows1, flat_base, N,
                       ep_return, ep_length, episode_stat);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_bootstrap_scatter(const int32_t* idx_list, const float* values, int32_t count, float gamma, float* rewards_flat,
                         vf_stream_t stream)
{
    if (!idx_list || !values || !rewards_flat || count < 0) return vf::fail(VF_EINVAL, "vf_bootstrap_scatter: bad argument");
    if (count == 0) return VF_OK;
    hipLaunchKernelGGL(vf::k_bootstrap_scatter, dim3(vf::blocks_for(count)), dim3(vf::kBlock), 0, vf::as_stream(stream), idx_list, values,
                       count, gamma, rewards_flat);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_ppo_update(const vf_mlp_desc* fwd, const vf_mlp_bwd_desc* bwd, const float* params, const float* packed,
                  const float* in0, const float* in1, const float* log_std, const float* action, const float* old_log_prob,
                  const float* adv, const float* ret, float* stats, int32_t M, const vf_ppo_loss_cfg* cfg, float* scratch,
                  vf_stream_t stream)
{
    if (!fwd || !params || !packed || !in0 || !log_std || !action || !old_log_prob || !adv || !ret || !cfg || !scratch || M <= 0)
        return vf::fail(VF_EINVAL, "vf_ppo_update: bad argument");
    if (fwd->n_layers < 1 || fwd->n_layers > VF_MLP_MAX_LAYERS) return vf::fail(VF_EINVAL, "vf_ppo_update: bad layer count");
    if (int rc = check_bwd_desc(bwd, "vf_ppo_update")) return rc;
    const int nwaves = (M + 31) / 32;        // = partial rows of loss statistics in scratch (16 floats each; any count: k_fold_stats strides over them)
    hipStream_t st = vf::as_stream(stream);
    const int rc = vf::ppo_update_chain_try(fwd, bwd, params, packed, in0, in1, log_std, action, old_log_prob, adv, ret, scratch, cfg, M, st);
    if (rc < 0) return rc;
    if (rc == 0) return vf::fail(VF_EUNSUPPORTED, "vf_ppo_update: the layer tables are not an instantiated network class");
    if (stats)      // else: the caller folds the partial rows (vf_mlp_weight_grad_sumsq loss_stats)
        hipLaunchKernelGGL(vf::k_fold_stats, dim3(1), dim3(1024), 0, st, scratch, nwaves, stats, cfg->d_log_std_out, cfg->stats_accum);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_sumsq(const float* x, int64_t n, float* out1, float* scratch, vf_stream_t stream)
{
    if (!x || !out1 || !scratch || n <= 0) return vf::fail(VF_EINVAL, "vf_sumsq: bad argument");
    hipStream_t st = vf::as_stream(stream);
    if (n <= (1 << 20)) {      // the flat gradient of the actor-critic MLP: one block, one launch
        hipLaunchKernelGGL(vf::k_sumsq_block, dim3(1), dim3(1024), 0, st, x, (long)n, out1);
        VF_HIP(hipGetLastError());
        return VF_OK;
    }
    double* part = reinterpret_cast<double*>(scratch);
    const int nblk = vf::grid_for(n, 256);
    hipLaunchKernelGGL(vf::k_sum2_partial, dim3(nblk), dim3(vf::kBlock), 0, st, x, (long)n, part);
    hipLaunchKernelGGL(vf::k_sum2_final, dim3(1), dim3(64), 0, st, part, nblk, (double*)nullptr, out1);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

int vf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* grad_sumsq,
                 const vf_adam_cfg* cfg, vf_stream_t stream)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq || !cfg || n <= 0 || cfg->step <= 0 ||
        (cfg->max_grad_norm > 0 && !grad_sumsq && !cfg->sumsq_partials))
        return vf::fail(VF_EINVAL, "vf_adam_step: bad argument");
    if ((cfg->pack_map == nullptr) != (cfg->packed == nullptr))
        return vf::fail(VF_EINVAL, "vf_adam_step: pack_map and packed must be given together");
    float bc1, bc2_sqrt;
    vf::adam_bias(*cfg, &bc1, &bc2_sqrt);
    hipLaunchKernelGGL(vf::k_adam, dim3(vf::grid_for(n, 1024)), dim3(vf::kBlock), 0, vf::as_stream(stream), param, grad, exp_avg,
                       exp_avg_sq, (long)n, grad_sumsq, *cfg, bc1, bc2_sqrt);
    VF_HIP(hipGetLastError());
    return VF_OK;
}

}  // extern "C"
