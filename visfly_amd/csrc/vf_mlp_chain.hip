// Register-chained actor-critic forward for gfx950: one wave64 carries 32 rows through the WHOLE network, the
// activations never leave its registers.
//
// Every layer is computed transposed, Y^T = W X^T, with v_mfma_f32_32x32x2_f32:
//     A operand = weights   A[i = n][k]      lane = n + 32 kk
//     B operand = X^T       B[k][j = m]      lane = m + 32 kk
//     C / D     = Y^T       D[i = n][j = m]  lane (m, h = lane >> 5) holds n = 4 h + (r & 3) + 8 (r >> 2), r = 0..15
// A lane of the accumulator therefore holds 16 features of ITS OWN row m -- exactly what the B operand of the next
// layer wants from that lane, provided step s of the next layer reduces over k = kidx(s, h) = the feature register
// r = s % 16 of tile s / 16 holds: kidx = 32 (s / 16) + 8 ((s % 16) / 4) + 4 h + (s % 4).  The reduction order
// is free (a sum), so the weights are packed once in that order (k_mlp_pack_weights, image at vf_mlp_layer.wr_off:
// one contiguous 1 KiB block = the A fragments of four consecutive steps of one 32-feature output tile, one float4
// per lane) and the accumulator registers are fed back as B operands untouched: no LDS, no barriers, no
// inter-wave traffic.  Bias + ReLU run on the accumulator registers; the copies the backward needs are stored from
// them (float4 = 4 consecutive features of a row).
//
// The layer shapes must be compile-time (register arrays cannot be indexed at run time): the kernel is a template
// over the network class of the reference's policies (utils/policies/extractors.py:578-592,662-678;
// policies.py:18-49): NB extractor branches of two ReLU layers, concatenated, then policy / value trunks of two
// ReLU layers with a 4-wide / 1-wide head.  vf_mlp_forward picks it when the layer table matches an instantiated
// shape and falls back to the LDS kernel (k_mlp_forward) otherwise.
#include "vf_common.hpp"

namespace vf {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct __attribute__((packed, aligned(4))) f32x4u {   // 4 consecutive floats at dword alignment (bias vectors)
    float x, y, z, w;
};

struct ChainIo {
    const float* in[2];
    float* mean;
    float* value;
};

// compile-time description of one layer of the chain
struct ChainLayer {
    int desc;      // index in vf_mlp_desc.layer
    int obs;       // >= 0: reads observation `obs` (natural k order), -1: reads activation tiles
    int in0, nin;  // first input tile, number of input tiles (obs: nin = number of 8-wide k groups)
    int out0, nout;
    int relu;
};

// NB branches (KA / KB = first-layer widths padded to 8), hidden widths in 32-feature tiles
template <int NB_, int KA_, int KB_, int E1_, int E2_, int P1_, int P2_, int V1_, int V2_>
struct ChainNet {
    static constexpr int NB = NB_, E1 = E1_, E2 = E2_, P1 = P1_, P2 = P2_, V1 = V1_, V2 = V2_;
    static constexpr int kin(int b) { return b == 0 ? KA_ : KB_; }
    static constexpr int n_layers = 2 * NB + 6;
    // tiles: [branch L1 outputs][feat][pi1][pi2][mean][vf1][vf2][value]
    static constexpr int t_e1(int b) { return b * E1; }
    static constexpr int t_feat = NB * E1;
    static constexpr int t_p1 = t_feat + NB * E2, t_p2 = t_p1 + P1, t_mean = t_p2 + P2;
    static constexpr int t_v1 = t_mean + 1, t_v2 = t_v1 + V1, t_val = t_v2 + V2;
    static constexpr int n_tiles = t_val + 1;
    // execution order alternates between independent chains so that one chain's epilogue (VALU) can sit in the
    // shadow of the other's MFMAs
    static constexpr ChainLayer layer(int i)
    {
        if (i < NB) return ChainLayer{2 * i, i, 0, kin(i) / 8, t_e1(i), E1, 1};
        if (i < 2 * NB) return ChainLayer{2 * (i - NB) + 1, -1, t_e1(i - NB), E1, t_feat + (i - NB) * E2, E2, 1};
        const int base = 2 * NB;
        switch (i - base) {
        case 0: return ChainLayer{base + 0, -1, t_feat, NB * E2, t_p1, P1, 1};
        case 1: return ChainLayer{base + 3, -1, t_feat, NB * E2, t_v1, V1, 1};
        case 2: return ChainLayer{base + 1, -1, t_p1, P1, t_p2, P2, 1};
        case 3: return ChainLayer{base + 4, -1, t_v1, V1, t_v2, V2, 1};
        case 4: return ChainLayer{base + 2, -1, t_p2, P2, t_mean, 1, 0};
        default: return ChainLayer{base + 5, -1, t_v2, V2, t_val, 1, 0};
        }
    }
    static constexpr int groups(int i) { return layer(i).obs >= 0 ? layer(i).nin : layer(i).nin * 4; }   // float4 k groups
    static constexpr int items(int i) { return groups(i) * layer(i).nout; }
    static constexpr int n_items()
    {
        int n = 0;
        for (int i = 0; i < n_layers; ++i) n += items(i);
        return n;
    }
    static constexpr int layer_of(int item)
    {
        int i = 0;
        while (item >= items(i)) { item -= items(i); ++i; }
        return i;
    }
    static constexpr int first_item(int li)
    {
        int n = 0;
        for (int i = 0; i < li; ++i) n += items(i);
        return n;
    }
};

constexpr int kChainDepth = 8;   // weight blocks in flight: 8 x 4 MFMAs x 64 cycles = 2 k cycles of cover

template <class N>
struct ChainState {
    f32x16 t[N::n_tiles];
    float x[2][16];              // observation fragments: x[b][s] = X[m][2 s + h] (K padded to <= 32)
    float4 ring[kChainDepth];
    float4 bias[4][4];           // bias of the layer in flight: [out tile][g] -> features 32 a + 8 g + 4 h .. + 3
};

struct ChainArgs {
    vf_mlp_desc d;
    const float* params;
    const float* packed;
    ChainIo io;
    int M;
};

template <class N, int I>
__device__ __forceinline__ float4 chain_load(const ChainArgs& g, int lane)
{
    constexpr int li = N::layer_of(I), local = I - N::first_item(li);
    constexpr ChainLayer L = N::layer(li);
    constexpr int G = N::groups(li), gq = local / L.nout, a = local % L.nout;
    const float4* img = reinterpret_cast<const float4*>(g.packed + g.d.layer[L.desc].wr_off);
    return img[(a * G + gq) * 64 + lane];
}

// widths are compile-time (hidden layers: whole tiles; heads: 4 / 1 features in lane half 0, q = 0), so the bias loads
// and the epilogue carry no guards
template <class N, int LI>
__device__ __forceinline__ void chain_bias_load(const ChainArgs& g, ChainState<N>& st, int h)
{
    constexpr ChainLayer L = N::layer(LI);
    const float* b = g.params + g.d.layer[L.desc].b_off;    // b_off is only dword aligned
    if constexpr (LI == N::n_layers - 1) {
        st.bias[0][0] = make_float4(b[0], 0.0f, 0.0f, 0.0f);
    } else if constexpr (LI == N::n_layers - 2) {
        const f32x4u v = *reinterpret_cast<const f32x4u*>(b);
        st.bias[0][0] = make_float4(v.x, v.y, v.z, v.w);
    } else {
#pragma unroll
        for (int a = 0; a < L.nout; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4u v = *reinterpret_cast<const f32x4u*>(b + 32 * a + 8 * q + 4 * h);
                st.bias[a][q] = make_float4(v.x, v.y, v.z, v.w);
            }
    }
}

template <class N, int LI>
__device__ __forceinline__ void chain_epilogue(const ChainArgs& g, ChainState<N>& st, int row, int h, bool live)
{
    constexpr ChainLayer L = N::layer(LI);
    if constexpr (LI >= N::n_layers - 2) {             // heads: mean (M,4) / value (M,1); only lane half 0 holds them
        const f32x16& y = st.t[L.out0];
        const float4 bq = st.bias[0][0];
        if (live && h == 0) {
            if constexpr (LI == N::n_layers - 2)
                *reinterpret_cast<float4*>(g.io.mean + (size_t)row * 4) = make_float4(y[0] + bq.x, y[1] + bq.y, y[2] + bq.z, y[3] + bq.w);
            else g.io.value[row] = y[0] + bq.x;
        }
    } else {
#pragma unroll
        for (int a = 0; a < L.nout; ++a) {
            f32x16& y = st.t[L.out0 + a];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = st.bias[a][q];
                y[4 * q + 0] += bq.x; y[4 * q + 1] += bq.y; y[4 * q + 2] += bq.z; y[4 * q + 3] += bq.w;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] = fmaxf(y[r], 0.0f);
        }
    }
}

// The copies of a layer's output that the backward reads are not stored in the epilogue (a 16 KiB burst per wave, all
// waves in lock-step, behind which the weight loads of the following items would queue: loads and stores retire in
// order on gfx9's vmcnt) but trickled out, a float4 or two per item of the NEXT layer in execution order.
template <class N, int LI, int LOCAL>
__device__ __forceinline__ void chain_deferred_store(const ChainArgs& g, const ChainState<N>& st, int row, int h, bool live)
{
    if constexpr (LI >= 1 && LI - 1 < N::n_layers - 2) {
        constexpr ChainLayer P = N::layer(LI - 1);
        constexpr int S = P.nout * 4, per = (S + N::items(LI) - 1) / N::items(LI);
        constexpr int s0 = LOCAL * per, s1 = (LOCAL + 1) * per < S ? (LOCAL + 1) * per : S;
        if constexpr (s0 < s1) {
            const vf_mlp_layer& D = g.d.layer[P.desc];
            if (D.save && live) {
                float* base = D.save + (size_t)row * D.save_ld + D.dst_col + 4 * h;
#pragma unroll
                for (int i = s0; i < s1; ++i) {
                    const int a = i / 4, q = i % 4;
                    const f32x16& y = st.t[P.out0 + a];
                    *reinterpret_cast<float4*>(base + 32 * a + 8 * q) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
                }
            }
        }
    }
}

template <class N, int I>
__device__ __forceinline__ void chain_items(const ChainArgs& g, ChainState<N>& st, int lane, int row, bool live)
{
    if constexpr (I < N::n_items()) {
        constexpr int li = N::layer_of(I), local = I - N::first_item(li);
        constexpr ChainLayer L = N::layer(li);
        constexpr int gq = local / L.nout, a = local % L.nout;
        const int h = lane >> 5;
        if constexpr (I + kChainDepth < N::n_items()) {
            const float4 nxt = chain_load<N, I + kChainDepth>(g, lane);
            // the slot being refilled is the one this item consumes: read it first
            const float4 w = st.ring[I % kChainDepth];
            st.ring[I % kChainDepth] = nxt;
            if constexpr (local == 0) chain_bias_load<N, li>(g, st, h);
            f32x16& acc = st.t[L.out0 + a];
            if constexpr (gq == 0) acc = f32x16{0};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float b;
                if constexpr (L.obs >= 0) b = st.x[L.obs][4 * gq + j];
                else b = st.t[L.in0 + gq / 4][4 * (gq % 4) + j];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(j == 0 ? w.x : j == 1 ? w.y : j == 2 ? w.z : w.w, b, acc, 0, 0, 0);
            }
        } else {
            const float4 w = st.ring[I % kChainDepth];
            if constexpr (local == 0) chain_bias_load<N, li>(g, st, h);
            f32x16& acc = st.t[L.out0 + a];
            if constexpr (gq == 0) acc = f32x16{0};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float b;
                if constexpr (L.obs >= 0) b = st.x[L.obs][4 * gq + j];
                else b = st.t[L.in0 + gq / 4][4 * (gq % 4) + j];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(j == 0 ? w.x : j == 1 ? w.y : j == 2 ? w.z : w.w, b, acc, 0, 0, 0);
            }
        }
        chain_deferred_store<N, li, local>(g, st, row, h, live);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (local == N::items(li) - 1) chain_epilogue<N, li>(g, st, row, h, live);
        chain_items<N, I + 1>(g, st, lane, row, live);
    }
}

template <class N, int I>
__device__ __forceinline__ void chain_prologue(const ChainArgs& g, ChainState<N>& st, int lane)
{
    if constexpr (I < kChainDepth && I < N::n_items()) {
        st.ring[I] = chain_load<N, I>(g, lane);
        chain_prologue<N, I + 1>(g, st, lane);
    }
}

template <class N>
__global__ __launch_bounds__(64) void k_mlp_forward_chain(const ChainArgs g)
{
    const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
    const int row = blockIdx.x * 32 + m;
    const bool live = row < g.M;
    const int rc = live ? row : g.M - 1;
    ChainState<N> st;
    chain_prologue<N, 0>(g, st, lane);
#pragma unroll
    for (int b = 0; b < N::NB; ++b) {
        const int w = g.d.in_dim[b];
        const float* x = g.io.in[b] + (size_t)rc * w;
#pragma unroll
        for (int s = 0; s < N::kin(b) / 2; ++s) {
            const int k = 2 * s + h;
            const float v = x[k < w ? k : w - 1];
            st.x[b][s] = k < w ? v : 0.0f;
        }
    }
    chain_items<N, 0>(g, st, lane, row, live);
}

using NetHover = ChainNet<1, 16, 8, 4, 2, 2, 2, 2, 2>;   // StateExtractor [128, 64], pi / vf [64, 64]
using NetNav = ChainNet<2, 16, 8, 4, 2, 2, 2, 2, 2>;     // StateTargetExtractor [128, 64] x 2, pi / vf [64, 64]

// does the layer table describe network class N (shapes, wiring, execution order of MlpPolicy)?
template <class N>
bool chain_matches(const vf_mlp_desc& d)
{
    if (d.n_layers != N::n_layers || d.n_inputs != N::NB) return false;
    for (int b = 0; b < N::NB; ++b)
        if (d.in_dim[b] < 1 || d.in_dim[b] > N::kin(b)) return false;
    auto is = [&](int li, int K, int No, int relu) {
        const vf_mlp_layer& L = d.layer[li];
        return L.K == K && L.No == No && (L.relu != 0) == (relu != 0) && L.wr_off >= 0 && (L.wr_off & 3) == 0;
    };
    const int feat = N::NB * N::E2 * 32;
    for (int b = 0; b < N::NB; ++b) {
        if (!is(2 * b, d.in_dim[b], N::E1 * 32, 1) || !is(2 * b + 1, N::E1 * 32, N::E2 * 32, 1)) return false;
        const vf_mlp_layer &l1 = d.layer[2 * b], &l2 = d.layer[2 * b + 1];
        if (l1.src != b || l1.src_col != 0 || l2.src != l1.dst || l2.src_col != l1.dst_col) return false;
        if (l2.dst_col != b * N::E2 * 32 || l2.dst != d.layer[1].dst) return false;
    }
    const int base = 2 * N::NB, fid = d.layer[1].dst;
    const int w1[2] = {N::P1 * 32, N::V1 * 32}, w2[2] = {N::P2 * 32, N::V2 * 32}, wo[2] = {4, 1};
    for (int t = 0; t < 2; ++t) {
        const int l = base + 3 * t;
        if (!is(l, feat, w1[t], 1) || !is(l + 1, w1[t], w2[t], 1) || !is(l + 2, w2[t], wo[t], 0)) return false;
        if (d.layer[l].src != fid || d.layer[l].src_col != 0) return false;
        if (d.layer[l + 1].src != d.layer[l].dst || d.layer[l + 2].src != d.layer[l + 1].dst) return false;
        if (d.layer[l + 2].dst != VF_MLP_OUT0 + t) return false;
    }
    for (int i = 0; i < d.n_layers; ++i) {
        const vf_mlp_layer& L = d.layer[i];
        if (L.save && ((L.save_ld & 3) || (L.dst_col & 3) || (reinterpret_cast<uintptr_t>(L.save) & 15))) return false;
    }
    return true;
}

template <class N>
int chain_launch(const vf_mlp_desc& d, const float* params, const float* packed, const float* in0, const float* in1, float* out0,
                 float* out1, int M, hipStream_t st)
{
    ChainArgs g{d, params, packed, ChainIo{{in0, in1}, out0, out1}, M};
    hipLaunchKernelGGL(k_mlp_forward_chain<N>, dim3((M + 31) / 32), dim3(64), 0, st, g);
    VF_HIP(hipGetLastError());
    return 1;
}

// 1: launched, 0: the layer table is not one of the instantiated network classes, < 0: error
int mlp_forward_chain_try(const vf_mlp_desc* d, const float* params, const float* packed, const float* in0, const float* in1,
                          float* out0, float* out1, int M, hipStream_t st)
{
    static const bool off = [] { const char* e = getenv("VISFLY_AMD_MLP_CHAIN"); return e && atoi(e) == 0; }();
    if (off || !out0 || !out1 || (reinterpret_cast<uintptr_t>(out0) & 15)) return 0;
    if (chain_matches<NetNav>(*d) && in1) return chain_launch<NetNav>(*d, params, packed, in0, in1, out0, out1, M, st);
    if (chain_matches<NetHover>(*d)) return chain_launch<NetHover>(*d, params, packed, in0, nullptr, out0, out1, M, st);
    return 0;
}

}  // namespace vf
