// vf_mlp_chain_bwd.hpp -- the reverse register chain (data gradients of the whole network, 32 rows per wave) as device code,
// shared by the kernels of vf_mlp_chain.hip and the persistent reverse sweep of vf_bptt_reverse.hip.
#pragma once
#include "vf_mlp_chain.hpp"

namespace vf {

// ------------------------------------------------------------------------------------------------
// Reverse chain: data gradients of the whole network for 32 rows per wave, again in registers.
// dA^T[k][m] = sum_n W[n][k] dZ^T[n][m] has the same shape as the forward step (A operand = W^T blocks in the order an
// accumulator lane holds its row's features: image at vf_mlp_layer.wq_off; B operand = the masked gradient tiles),
// so the gradient tiles are fed back exactly like the activations.  After a layer's gradient w.r.t. the previous
// activation is complete it is masked with that activation (> 0, read back from the copy the forward saved) and
// stored to the buffer the weight-gradient kernel (vf_mlp_wgrad.hip) reads as dZ; the stores trickle out through the
// items of the next op like the forward's.  The weight / bias gradients are NOT formed here: they reduce over rows,
// i.e. across waves, and have their own kernel with row-slab partials.
// ------------------------------------------------------------------------------------------------
struct BwdFin {     // mask tiles [t0, t0 + nt) with the saved output of forward layer fl and store them as its dZ
    int fl, t0, nt, ym0;
    // slices (vf_mlp_chain_split.hpp); defaults = the whole layer
    int c0 = 0;            // the tiles are tiles [c0, c0 + nt) of layer fl's output (mask source, dZ columns)
    int s0 = 0, sn = -1;   // of those, [s0, s0 + sn) are stored by THIS wave (-1: all)
};
struct BwdOp {
    int fl;         // forward layer whose weights are applied (MlpPolicy order)
    int in_kind;    // 0: gradient tiles, 1: d_mean (M,4), 2: d_value (M,)
    int in0, G;     // first input tile, number of 8-feature groups of the input gradient
    int out0, nout; // output tiles = ceil(K / 32)
    int accum;      // 1: keep accumulating into the output tiles
    int obs;        // >= 0: the output is dLoss/d observation `obs` (stored directly, no mask)
    int nfin;
    BwdFin fin[2];
    // slices (vf_mlp_chain_split.hpp); defaults = the whole op
    int a0 = 0;            // the op's output tiles [a0, a0 + nout) are computed here (weight image)
    int xsn = 0, xs0 = 0, xr0 = 0;   // > 0: before the op is finalised, tiles [xs0, xs0 + xsn) go to the partner wave through LDS and its xsn tiles are ADDED to [xr0, ..)
};

template <class N, bool PI, bool VF, bool IG>
struct BwdProg {
    using Net = N;
    static constexpr int NB = N::NB;
    static constexpr bool sac_head = PI && VF && N::HV == 4 && N::HM == 4;     // td_policies.Actor: mu / log_std heads of ONE action
    static constexpr int L_pi0 = 2 * NB, L_pi1 = 2 * NB + 1, L_mean = 2 * NB + 2, L_vf0 = 2 * NB + 3, L_vf1 = 2 * NB + 4, L_val = 2 * NB + 5;
    // gradient tiles
    static constexpr int g_p2 = 0, g_v2 = g_p2 + N::P2, g_p1 = g_v2 + N::V2, g_v1 = g_p1 + N::P1, g_feat = g_v1 + N::V1;
    static constexpr int g_e1(int b) { return g_feat + NB * N::E2 + b * N::E1; }
    static constexpr int g_in(int b) { return g_e1(NB) + b; }
    static constexpr int n_tiles = g_in(NB);
    static constexpr int n_ym = 4;          // BwdFin::ym0 + nt of every op fits: mask slots of the stand-alone reverse chain
    static constexpr int n_ops = (PI ? 3 : 0) + (VF ? 3 : 0) + NB + (IG ? NB : 0);
    static constexpr bool included(int l) { return l < 2 * NB || (l >= L_pi0 && l <= L_mean && PI) || (l >= L_vf0 && VF); }
    static constexpr int entry(int fl)      // index of forward layer fl in the (reversed, trunk-skipping) vf_mlp_bwd_desc
    {
        int e = 0;
        for (int l = fl + 1; l < 2 * NB + 6; ++l) e += included(l) ? 1 : 0;
        return e;
    }
    static constexpr BwdOp feat_fins(BwdOp o)
    {
        o.nfin = NB;
        for (int b = 0; b < NB; ++b) o.fin[b] = BwdFin{2 * b + 1, g_feat + b * N::E2, N::E2, b * N::E2};
        return o;
    }
    static constexpr BwdOp op(int i)
    {
        // order: heads, second trunk layers, first trunk layers (-> feat), extractor L2 layers, [extractor L1 layers]
        int k = 0;
        if (PI) { if (i == k) return BwdOp{L_mean, 1, 0, 1, g_p2, N::P2, 0, -1, 1, {BwdFin{L_pi1, g_p2, N::P2, 0}, {}}}; ++k; }
        if (VF) { if (i == k) return BwdOp{L_val, 2, 0, 1, g_v2, N::V2, 0, -1, 1, {BwdFin{L_vf1, g_v2, N::V2, 2}, {}}}; ++k; }
        if (PI) { if (i == k) return BwdOp{L_pi1, 0, g_p2, N::P2 * 4, g_p1, N::P1, 0, -1, 1, {BwdFin{L_pi0, g_p1, N::P1, 0}, {}}}; ++k; }
        if (VF) { if (i == k) return BwdOp{L_vf1, 0, g_v2, N::V2 * 4, g_v1, N::V1, 0, -1, 1, {BwdFin{L_vf0, g_v1, N::V1, 2}, {}}}; ++k; }
        if (PI) {
            if (i == k) {
                BwdOp o{L_pi0, 0, g_p1, N::P1 * 4, g_feat, NB * N::E2, 0, -1, 0, {}};
                return VF ? o : feat_fins(o);
            }
            ++k;
        }
        if (VF) {
            if (i == k) return feat_fins(BwdOp{L_vf0, 0, g_v1, N::V1 * 4, g_feat, NB * N::E2, PI ? 1 : 0, -1, 0, {}});
            ++k;
        }
        for (int b = 0; b < NB; ++b) {
            if (i == k) return BwdOp{2 * b + 1, 0, g_feat + b * N::E2, N::E2 * 4, g_e1(b), N::E1, 0, -1, 1, {BwdFin{2 * b, g_e1(b), N::E1, 0}, {}}};
            ++k;
        }
        for (int b = 0; b < NB; ++b) {
            if (i == k) return BwdOp{2 * b, 0, g_e1(b), N::E1 * 4, g_in(b), 1, 0, b, 0, {}};
            ++k;
        }
        return BwdOp{};
    }
    static constexpr int items(int i) { return op(i).G * op(i).nout; }
    static constexpr int n_items()
    {
        int n = 0;
        for (int i = 0; i < n_ops; ++i) n += items(i);
        return n;
    }
    static constexpr int op_of(int item)
    {
        int i = 0;
        while (item >= items(i)) { item -= items(i); ++i; }
        return i;
    }
    static constexpr int first_item(int oi)
    {
        int n = 0;
        for (int i = 0; i < oi; ++i) n += items(i);
        return n;
    }
};

struct BwdArgsChain {
    vf_mlp_bwd_desc d;
    const float* packed;
    int M;
    // optional action head (vf_mlp_backward_data_act): the head gradient is formed here from d_action
    const float4* rp_d_action;
    const float4* rp_action;
    const float* rp_log_std;
    const float4* rp_eps;
    float4* rp_g_log_std;
    // sac_head classes: the log_std rows the forward saved (its second head; rp_log_std / rp_g_log_std unused) and the clamp bounds of
    // action = tanh(mean + eps exp(clamp(log_std, lo, hi))); both head gradients are formed from d_action (k_shac_head_bwd's arithmetic)
    const float4* rp_ls_rows;
    float rp_ls_lo, rp_ls_hi;
    // a persistent caller's exp(rp_log_std[k]), computed once per launch (rp_std_valid != 0), instead of four loads + expf per pass
    int rp_std_valid;
    float rp_std[4];
};

// d log_std of one component: torch.clamp passes the gradient on the closed interval (k_shac_head_bwd)
__device__ __forceinline__ float sac_dls(float dp, float e, float s, float lo, float hi) { return (s >= lo && s <= hi) ? dp * e * expf(s) : 0.0f; }

template <class P>
struct BwdState {
    f32x16 t[P::n_tiles];
    float4 ring[kChainDepth];
    float4 ym[P::n_ym][4];       // saved activations (mask source) of the tiles being finalised
    float hin[2][4];             // head gradients of this lane's row: d_mean[0..3] / d_value (lane half 0), else 0
    unsigned long sv_base[2];    // dZ buffers of the (up to two) layers the PREVIOUS op finalised (bwd_store_setup): uniform bases ...
    unsigned sv_off[2];          // ... + this lane's byte offsets
};

template <class P, int I>
__device__ __forceinline__ float4 bwd_load(const BwdArgsChain& g, int lane)
{
    constexpr int oi = P::op_of(I), local = I - P::first_item(oi);
    constexpr BwdOp O = P::op(oi);
    constexpr int gq = local / O.nout, a = local % O.nout;
#if VF_CHAIN_BUFFER_LOADS
    return chain_buffer_float4(chain_weight_rsrc(g.packed), (unsigned)lane * 16u, (unsigned)g.d.layer[P::entry(O.fl)].wq_off * 4u + ((O.a0 + a) * O.G + gq) * 1024u);
#else
    const char* base = reinterpret_cast<const char*>(g.packed + g.d.layer[P::entry(O.fl)].wq_off) + ((O.a0 + a) * O.G + gq) * 1024;
    return *reinterpret_cast<const float4*>(base + (unsigned)lane * 16u);
#endif
}

template <class P, int OI>
__device__ __forceinline__ void bwd_mask_load(const BwdArgsChain& g, BwdState<P>& st, int rc, int h)
{
    constexpr BwdOp O = P::op(OI);
#pragma unroll
    for (int f = 0; f < O.nfin; ++f) {
        const vf_mlp_bwd_layer& E = g.d.layer[P::entry(O.fin[f].fl)];
        const float* y = E.Y + (size_t)rc * E.ld_y + 4 * h;
#pragma unroll
        for (int a = 0; a < O.fin[f].nt; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) st.ym[O.fin[f].ym0 + a][q] = *reinterpret_cast<const float4*>(y + 32 * (O.fin[f].c0 + a) + 8 * q);
    }
}

struct NoFwd {};     // FS of the stand-alone reverse chain: masks are read back from HBM

// ReLU mask of the tiles of fin F of op OI: from the saved copy (NoFwd), from the fused kernel's own forward tiles, or from what a
// fused kernel kept of them -- one bit per value (ChainLayer::pk0)
template <class P, class FS, int OI, int F>
__device__ __forceinline__ void bwd_mask_fin(BwdState<P>& st, const FS& fs)
{
    constexpr BwdOp O = P::op(OI);
    if constexpr (F < O.nfin) {
        constexpr BwdFin fin = O.fin[F];
        constexpr int bits0 = [] { if constexpr (std::is_same<FS, NoFwd>::value) return -1; else return FS::Net::mask_bits(fin.fl); }();
#pragma unroll
        for (int a = 0; a < fin.nt; ++a) {
            f32x16& v = st.t[fin.t0 + a];
            if constexpr (bits0 >= 0) {
                const int j = bits0 + fin.c0 + a;
                const unsigned w = fs.mb[j >> 1];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int keep = (int)(w << (31 - (16 * (j & 1) + r))) >> 31;      // 0 / -1
                    const float vr = v[r];      // (a copy: see chain_pack_input)
                    v[r] = __uint_as_float(__float_as_uint(vr) & (unsigned)keep);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 y;
                    if constexpr (std::is_same<FS, NoFwd>::value) y = st.ym[fin.ym0 + a][q];
                    else {      // fused kernel: the forward's own accumulator tile of that layer is still in registers
                        const f32x16& t = fs.t[P::Net::tile_of_layer(fin.fl) + fin.c0 + a];
                        y = make_float4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
                    }
                    constexpr int kind = P::Net::act_of(fin.fl);      // d/dz from the layer's OUTPUT (vf_common.hpp: act_mul)
                    v[4 * q + 0] = act_mul_c<kind>(v[4 * q + 0], y.x);
                    v[4 * q + 1] = act_mul_c<kind>(v[4 * q + 1], y.y);
                    v[4 * q + 2] = act_mul_c<kind>(v[4 * q + 2], y.z);
                    v[4 * q + 3] = act_mul_c<kind>(v[4 * q + 3], y.w);
                }
            }
        }
    }
}

template <class P, class FS, int OI>
__device__ __forceinline__ void bwd_finalize(const BwdArgsChain& g, BwdState<P>& st, const FS& fs, int row, int h, bool live)
{
    constexpr BwdOp O = P::op(OI);
    bwd_mask_fin<P, FS, OI, 0>(st, fs);
    bwd_mask_fin<P, FS, OI, 1>(st, fs);
    if constexpr (O.obs >= 0) {        // dLoss/d observation: features 4 h + (r & 3) + 8 (r >> 2) of this lane's row
        const vf_mlp_bwd_layer& E = g.d.layer[P::entry(O.fl)];
        if (live) {
            const f32x16& v = st.t[O.out0];
            float* dx = E.dX + (size_t)row * E.ld_dx;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = 4 * h + (r & 3) + 8 * (r >> 2);
                if (k < E.K) dx[k] = v[r];
            }
        }
    }
}

constexpr int bwd_fin_stored(const BwdFin& f) { return f.sn < 0 ? f.nt : f.sn; }

// stores of the tiles the PREVIOUS op finalised, spread over this op's items.  Base pointers / row strides are read from the layer
// table once per op and pinned (chain_store_setup, vf_mlp_chain.hpp, says why); lanes past the last row replicate row M - 1: no guard
template <class P, int OI>
__device__ __forceinline__ void bwd_store_setup(const BwdArgsChain& g, BwdState<P>& st, int rc, int h)
{
    if constexpr (OI >= 1) {
        constexpr BwdOp Q = P::op(OI - 1);
#pragma unroll
        for (int f = 0; f < Q.nfin; ++f) {
            const vf_mlp_bwd_layer& E = g.d.layer[P::entry(Q.fin[f].fl)];
            unsigned long b = reinterpret_cast<unsigned long>(E.dY);
            asm volatile("" : "+s"(b));
            st.sv_base[f] = b;
            st.sv_off[f] = ((unsigned)rc * (unsigned)E.ld_dy + 4u * h) * 4u;                    // bytes
        }
    }
}

template <class P, int OI, int LOCAL>
__device__ __forceinline__ void bwd_deferred_store(const BwdState<P>& st)
{
    if constexpr (OI >= 1) {
        constexpr BwdOp Q = P::op(OI - 1);
        constexpr int S0 = Q.nfin > 0 ? bwd_fin_stored(Q.fin[0]) * 4 : 0, S = S0 + (Q.nfin > 1 ? bwd_fin_stored(Q.fin[1]) * 4 : 0);
        constexpr int n_it = P::items(OI), per = (S + n_it - 1) / n_it;
        constexpr int s0 = LOCAL * per, s1 = (LOCAL + 1) * per < S ? (LOCAL + 1) * per : S;
        if constexpr (s0 < s1) {
#pragma unroll
            for (int i = s0; i < s1; ++i) {
                const int f = i < S0 ? 0 : 1, ii = i - (f ? S0 : 0), a = Q.fin[f].s0 + ii / 4, q = ii % 4;
                const f32x16& v = st.t[Q.fin[f].t0 + a];
                chain_buffer_store(chain_store_rsrc(st.sv_base[f]), st.sv_off[f], (32 * (Q.fin[f].c0 + a) + 8 * q) * 4, v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            }
        }
    }
}

// partial gradients of the two halves of a split row tile meet (vf_mlp_chain_split.hpp): send, barrier, add the partner's
template <class P, int OI>
__device__ __forceinline__ void bwd_exchange(BwdState<P>& st, int lane)
{
    constexpr BwdOp O = P::op(OI);
    if constexpr (O.xsn > 0) {
        static_assert(O.xsn <= kXchTiles, "exchange buffer");
        constexpr int R = P::role;
#pragma unroll
        for (int a = 0; a < O.xsn; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x16& v = st.t[O.xs0 + a];
                vf_xch_bwd[R][a][q][lane] = vf_st4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            }
        xch_barrier();
#pragma unroll
        for (int a = 0; a < O.xsn; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const vf_st4 u = vf_xch_bwd[1 - R][a][q][lane];
                f32x16& v = st.t[O.xr0 + a];
                v[4 * q] += u[0]; v[4 * q + 1] += u[1]; v[4 * q + 2] += u[2]; v[4 * q + 3] += u[3];
            }
    }
}

template <class P, class FS, int I>
__device__ __forceinline__ void bwd_items(const BwdArgsChain& g, BwdState<P>& st, const FS& fs, int lane, int row, int rc, bool live)
{
    if constexpr (I < P::n_items()) {
        constexpr int oi = P::op_of(I), local = I - P::first_item(oi);
        constexpr BwdOp O = P::op(oi);
        constexpr int gq = local / O.nout, a = local % O.nout;
        const int h = lane >> 5;
        const float4 w = st.ring[I % kChainDepth];
        if constexpr (I + kChainDepth < P::n_items()) st.ring[I % kChainDepth] = bwd_load<P, I + kChainDepth>(g, lane);
        if constexpr (local == 0 && O.in_kind == 0 && std::is_same<FS, NoFwd>::value) bwd_mask_load<P, oi>(g, st, rc, h);   // head ops: in the prologue
        f32x16& acc = st.t[O.out0 + a];
        if constexpr (gq == 0 && !O.accum) acc = f32x16{0};
        __builtin_amdgcn_sched_barrier(0);      // refill load + address arithmetic before the item's MFMAs, not among them (chain_items)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float b;
            if constexpr (O.in_kind == 0) b = st.t[O.in0 + gq / 4][4 * (gq % 4) + j];
            else b = st.hin[O.in_kind - 1][j];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(j == 0 ? w.x : j == 1 ? w.y : j == 2 ? w.z : w.w, b, acc, 0, 0, 0);
        }
        if constexpr (local == 0) bwd_store_setup<P, oi>(g, st, rc, h);
        bwd_deferred_store<P, oi, local>(st);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (local == P::items(oi) - 1) {
            VF_CHAIN_HOOK(2, oi);
            bwd_exchange<P, oi>(st, lane);
            bwd_finalize<P, FS, oi>(g, st, fs, row, h, live);
            VF_CHAIN_HOOK(3, oi);
        }
        bwd_items<P, FS, I + 1>(g, st, fs, lane, row, rc, live);
    }
}

template <class P, int I>
__device__ __forceinline__ void bwd_prologue(const BwdArgsChain& g, BwdState<P>& st, int lane)
{
    if constexpr (I < kChainDepth && I < P::n_items()) {
        st.ring[I] = bwd_load<P, I>(g, lane);
        bwd_prologue<P, I + 1>(g, st, lane);
    }
}

template <class P, int OI>
__device__ __forceinline__ void bwd_head_prologue(const BwdArgsChain& g, BwdState<P>& st, int rc, int h, bool live)
{
    if constexpr (OI < P::n_ops) {
        constexpr BwdOp O = P::op(OI);
        if constexpr (O.in_kind != 0) {
            const vf_mlp_bwd_layer& E = g.d.layer[P::entry(O.fl)];
            const float* dy = E.dY + (size_t)rc * E.ld_dy;
            if constexpr (O.in_kind == 1) {
                if (g.rp_d_action) {       // k_reparam_bwd's arithmetic; lane half 0 of a live row writes d_mean / g_log_std
                    const float4 da = g.rp_d_action[rc], a = g.rp_action[rc], e = g.rp_eps[rc];
                    const float4 dm = make_float4(da.x * (1.0f - a.x * a.x), da.y * (1.0f - a.y * a.y), da.z * (1.0f - a.z * a.z),
                                                  da.w * (1.0f - a.w * a.w));
                    if constexpr (P::sac_head) {       // ... k_shac_head_bwd's: d_mu = dm, d_log_std (the second head's gradient) from the saved row
                        const vf_mlp_bwd_layer& EV = g.d.layer[P::entry(P::L_val)];
                        const float4 s = g.rp_ls_rows[rc];
                        const float lo = g.rp_ls_lo, hi = g.rp_ls_hi;
                        const float4 dl = make_float4(sac_dls(dm.x, e.x, s.x, lo, hi), sac_dls(dm.y, e.y, s.y, lo, hi),
                                                      sac_dls(dm.z, e.z, s.z, lo, hi), sac_dls(dm.w, e.w, s.w, lo, hi));
                        if (live && h == 0) {
                            *reinterpret_cast<float4*>(const_cast<float*>(E.dY) + (size_t)rc * E.ld_dy) = dm;
                            *reinterpret_cast<float4*>(const_cast<float*>(EV.dY) + (size_t)rc * EV.ld_dy) = dl;
                        }
                        st.hin[1][0] = h == 0 ? dl.x : 0.0f; st.hin[1][1] = h == 0 ? dl.y : 0.0f;
                        st.hin[1][2] = h == 0 ? dl.z : 0.0f; st.hin[1][3] = h == 0 ? dl.w : 0.0f;
                    } else if (live && h == 0) {
                        *reinterpret_cast<float4*>(const_cast<float*>(E.dY) + (size_t)rc * E.ld_dy) = dm;
                        float4 gl = g.rp_g_log_std[rc];
                        gl.x += dm.x * expf(g.rp_log_std[0]) * e.x; gl.y += dm.y * expf(g.rp_log_std[1]) * e.y;
                        gl.z += dm.z * expf(g.rp_log_std[2]) * e.z; gl.w += dm.w * expf(g.rp_log_std[3]) * e.w;
                        g.rp_g_log_std[rc] = gl;
                    }
                    st.hin[0][0] = h == 0 ? dm.x : 0.0f; st.hin[0][1] = h == 0 ? dm.y : 0.0f;
                    st.hin[0][2] = h == 0 ? dm.z : 0.0f; st.hin[0][3] = h == 0 ? dm.w : 0.0f;
                } else if constexpr (P::Net::HM == 1) {      // first head 1 wide (the twin critic's Q1): d_q (M,)
                    st.hin[0][0] = h == 0 ? dy[0] : 0.0f; st.hin[0][1] = 0.0f; st.hin[0][2] = 0.0f; st.hin[0][3] = 0.0f;
                } else {
                    st.hin[0][0] = h == 0 ? dy[0] : 0.0f; st.hin[0][1] = h == 0 ? dy[1] : 0.0f;
                    st.hin[0][2] = h == 0 ? dy[2] : 0.0f; st.hin[0][3] = h == 0 ? dy[3] : 0.0f;
                }
            } else if constexpr (P::Net::HV == 4) {      // second head 4 wide (the SAC actor's log_std head): d_log_std (M,4)
                if (!(P::sac_head && g.rp_d_action)) {   // (else: formed by the first head's branch above)
                    st.hin[1][0] = h == 0 ? dy[0] : 0.0f; st.hin[1][1] = h == 0 ? dy[1] : 0.0f;
                    st.hin[1][2] = h == 0 ? dy[2] : 0.0f; st.hin[1][3] = h == 0 ? dy[3] : 0.0f;
                }
            } else {
                st.hin[1][0] = h == 0 ? dy[0] : 0.0f; st.hin[1][1] = 0.0f; st.hin[1][2] = 0.0f; st.hin[1][3] = 0.0f;
            }
            bwd_mask_load<P, OI>(g, st, rc, h);
            bwd_head_prologue<P, OI + 1>(g, st, rc, h, live);
        }
    }
}

// the last op's finalised tiles have no following items to carry their stores
template <class P>
__device__ __forceinline__ void bwd_tail_store(const BwdArgsChain& g, const BwdState<P>& st, int row, int h, bool live)
{
    constexpr BwdOp Q = P::op(P::n_ops - 1);
    if (!live) return;
#pragma unroll
    for (int f = 0; f < Q.nfin; ++f) {
        const vf_mlp_bwd_layer& E = g.d.layer[P::entry(Q.fin[f].fl)];
        float* base = const_cast<float*>(E.dY) + (size_t)row * E.ld_dy + 4 * h;
#pragma unroll
        for (int i = 0; i < bwd_fin_stored(Q.fin[f]); ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int a = Q.fin[f].s0 + i;
                const f32x16& v = st.t[Q.fin[f].t0 + a];
                *reinterpret_cast<float4*>(base + 32 * (Q.fin[f].c0 + a) + 8 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// The same reverse chain with 16 rows per wave (v_mfma_f32_16x16x4_f32) for SMALL row counts, the counterpart of the 16-row
// forward (vf_mlp_chain.hpp): at the BPTT shard (16 384 rows) the 32-row reverse chain puts 512 waves on 1 024 SIMDs and lasts as
// long as one wave needs for the whole network; 16 rows per wave halve that.
//     A operand = W^T     A[i = k][kk = n]   lane = k + 16 kq
//     B operand = dZ^T    B[kk = n][j = m]   lane = m + 16 kq
//     C / D     = dX^T    D[i = k][j = m]    lane (m, gq = lane >> 4) holds k = 4 gq + r, r = 0..3
// A gradient-tile lane holds features 4 gq + r of its own row; as the B operand of step r it supplies n = 16 T + 4 kq + r, so step
// r's A fragment is W[16 T + 4 kq + r][16 a + i]: the four steps of an (input tile T, output tile a) item are one float4 of the
// 32-row reverse chain's image (vf_mlp_layer.wq_off, bwd16_load), in lane order; no image of its own.  (Four dwords of the
// zero-padded row-major data-gradient image, vf_mlp_layer.wb_off, are the A/B alternative: same time.)  Heads (4 / 1 input features) are ONE MFMA per output tile: B = the lane's kq-th head gradient.
// Summation order inside a dot product differs from the 32-row chain's (4 products per MFMA instead of 2), so the two agree to
// rounding, not to the bit: callers pick ONE of them per row count (bwd16_ok, the forward's rule) on every path.
// ------------------------------------------------------------------------------------------------
template <class P>
struct Bwd16 {
    static constexpr int nin(int oi) { return P::op(oi).in_kind == 0 ? P::op(oi).G / 2 : 1; }       // 16-feature input tiles
    static constexpr int nout(int oi) { return P::op(oi).obs >= 0 ? 1 : 2 * P::op(oi).nout; }       // 16-feature output tiles
    static constexpr int k32(int oi) { return P::op(oi).obs >= 0 ? 32 : 32 * P::op(oi).nout; }      // row length of the wb image
    static constexpr int items(int oi) { return nin(oi) * nout(oi); }
    // mask float4s (saved activations of the 16-tiles an op finalises) live in ONE register array, op oi's at mask0(oi) ..
    static constexpr int nmask(int oi)
    {
        int n = 0;
        for (int f = 0; f < P::op(oi).nfin; ++f) n += 2 * P::op(oi).fin[f].nt;
        return n;
    }
    static constexpr int mask0(int oi)
    {
        int n = 0;
        for (int i = 0; i < oi; ++i) n += nmask(i);
        return n;
    }
    static constexpr int n_masks() { return mask0(P::n_ops); }
    static constexpr int n_items()
    {
        int n = 0;
        for (int i = 0; i < P::n_ops; ++i) n += items(i);
        return n;
    }
    static constexpr int op_of(int item)
    {
        int i = 0;
        while (item >= items(i)) { item -= items(i); ++i; }
        return i;
    }
    static constexpr int first_item(int oi)
    {
        int n = 0;
        for (int i = 0; i < oi; ++i) n += items(i);
        return n;
    }
};

template <class P>
struct BwdState16 {
    f32x4 t[2 * P::n_tiles];
    float4 ring[kChain16Depth];  // the four A fragments of an item
    float4 ym[Bwd16<P>::n_masks() > 0 ? Bwd16<P>::n_masks() : 1];   // saved activations (mask source) of the 16-tiles every op finalises
    float hin[2];                // this lane's element of the head gradients: d_mean[kq] / d_value (kq = 0), else 0
    float4 pa, pe, pg;           // preloaded action / noise / log_std-gradient rows of the action head's reverse (bwd16_mask_preload)
    float4 pda;                  // PRE callers: the row's d_action (k_bptt_reverse hands it over through LDS, not through rp_d_action)
    vf_gptr sv_base[2];      // dZ buffers of the layers the previous op finalised (bwd16_store_setup) + this lane's byte offsets
    unsigned sv_off[2];
};

template <class P, int I>
__device__ __forceinline__ float4 bwd16_load(const BwdArgsChain& g, int lane)
{
    using B = Bwd16<P>;
    constexpr int oi = B::op_of(I), local = I - B::first_item(oi);
    constexpr BwdOp O = P::op(oi);
    constexpr int T = local / B::nout(oi), a = local % B::nout(oi), K32 = B::k32(oi);
#if VF_CHAIN16_WT == 2
    {   // ONE float4 of the 32-row reverse chain's own image (vf_mlp_layer.wq_off, block (a, g) = 1 KiB, lane l = k + 32 h holds
        // W[32 (g >> 2) + 8 (g & 3) + 4 h .. + 3][32 a + k]): the fragment of lane (i, kq) for output 16-tile a16, input 16-tile T is
        // the float4 of lane 16 (a16 & 1) + i + 32 (kq & 1) in block (a16 / 2, 4 (T >> 1) + 2 (T & 1) + (kq >> 1)); quarter-waves of
        // 256 contiguous bytes, one instruction per item (chain16_load).  Heads: component kq of lane k's float4 in block (a, 0)
        constexpr int GQ = O.in_kind == 0 ? O.G : 1;
        constexpr int blk = (a >> 1) * GQ + (O.in_kind == 0 ? 4 * (T >> 1) + 2 * (T & 1) : 0);
        const unsigned kq = lane >> 4;
        const char* qb = reinterpret_cast<const char*>(g.packed + g.d.layer[P::entry(O.fl)].wq_off) + (blk * 1024 + 256 * (a & 1));
        if constexpr (O.in_kind == 0)
            return *reinterpret_cast<const float4*>(qb + ((kq >> 1) * 1024u + (kq & 1u) * 512u + (unsigned)(lane & 15) * 16u));
        else
            return make_float4(*reinterpret_cast<const float*>(qb + ((unsigned)(lane & 15) * 16u + kq * 4u)), 0.0f, 0.0f, 0.0f);
    }
#endif
    const float* base = g.packed + g.d.layer[P::entry(O.fl)].wb_off;             // wave-uniform
    if constexpr (O.in_kind == 0) {
        const float* p = base + (16 * T * K32 + 16 * a) + ((unsigned)(lane >> 4) * (4u * K32) + (unsigned)(lane & 15));
        return make_float4(p[0], p[K32], p[2 * K32], p[3 * K32]);
    } else {
        const float* p = base + 16 * a + ((unsigned)(lane >> 4) * (unsigned)K32 + (unsigned)(lane & 15));
        return make_float4(p[0], 0.0f, 0.0f, 0.0f);
    }
}

template <class P, int OI>
__device__ __forceinline__ void bwd16_mask_load(const BwdArgsChain& g, BwdState16<P>& st, int rc, int gq)
{
    constexpr BwdOp O = P::op(OI);
    constexpr int m0 = Bwd16<P>::mask0(OI);
#pragma unroll
    for (int f = 0; f < O.nfin; ++f) {
        const vf_mlp_bwd_layer& E = g.d.layer[P::entry(O.fin[f].fl)];
        const float* y = E.Y + (size_t)rc * E.ld_y + 4 * gq;
        const int f0 = f == 0 ? 0 : 2 * O.fin[0].nt;
#pragma unroll
        for (int a = 0; a < 2 * O.fin[f].nt; ++a) st.ym[m0 + f0 + a] = *reinterpret_cast<const float4*>(y + 16 * a);
    }
}

// every op's masks at once (a persistent sweep issues them ahead of the step's adjoint: the saved activations of a slot were
// written a whole forward sweep ago, HBM by now, and an op is over long before its own loads would be back)
template <class P, int OI>
__device__ __forceinline__ void bwd16_mask_preload(const BwdArgsChain& g, BwdState16<P>& st, int rc, int gq)
{
    if constexpr (OI == 0) {       // ... and the rows of the action head's reverse that do not depend on the adjoint
        if (g.rp_d_action) {
            st.pa = g.rp_action[rc];
            st.pe = g.rp_eps[rc];
            if constexpr (P::sac_head) st.pg = g.rp_ls_rows[rc];
            else st.pg = g.rp_g_log_std[rc];
        }
    }
    if constexpr (OI < P::n_ops) {
        bwd16_mask_load<P, OI>(g, st, rc, gq);
        bwd16_mask_preload<P, OI + 1>(g, st, rc, gq);
    }
}

template <class P, int OI>
__device__ __forceinline__ void bwd16_finalize(const BwdArgsChain& g, BwdState16<P>& st, int row, int gq, bool live)
{
    constexpr BwdOp O = P::op(OI);
    constexpr int m0 = Bwd16<P>::mask0(OI);     // (constexpr local: called in a runtime expression, the op table is walked at run time -- r04: the two-trunk Nav class)
#pragma unroll
    for (int f = 0; f < O.nfin; ++f)
#pragma unroll
        for (int a = 0; a < 2 * O.fin[f].nt; ++a) {
            f32x4& v = st.t[2 * O.fin[f].t0 + a];
            const float4 y = st.ym[m0 + (f == 0 ? 0 : 2 * O.fin[0].nt) + a];
            const int kind = P::Net::act_of(O.fin[f].fl);      // (a constant once the loop over f is unrolled: the switch folds away)
            v[0] = act_mul(v[0], y.x, kind);
            v[1] = act_mul(v[1], y.y, kind);
            v[2] = act_mul(v[2], y.z, kind);
            v[3] = act_mul(v[3], y.w, kind);
        }
    if constexpr (O.obs >= 0) {        // dLoss/d observation: features 4 gq + r of this lane's row
        const vf_mlp_bwd_layer& E = g.d.layer[P::entry(O.fl)];
        if (live) {
            const f32x4& v = st.t[2 * O.out0];
            float* dx = E.dX + (size_t)row * E.ld_dx;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * gq + r;
                if (k < E.K) dx[k] = v[r];
            }
        }
    }
}

// stores of the tiles the PREVIOUS op finalised, spread over this op's items (bases pinned per op: bwd_store_setup)
template <class P, int OI>
__device__ __forceinline__ void bwd16_store_setup(const BwdArgsChain& g, BwdState16<P>& st, int rc, int gq)
{
    if constexpr (OI >= 1) {
        constexpr BwdOp Q = P::op(OI - 1);
#pragma unroll
        for (int f = 0; f < Q.nfin; ++f) {
            const vf_mlp_bwd_layer& E = g.d.layer[P::entry(Q.fin[f].fl)];
            unsigned long b = reinterpret_cast<unsigned long>(E.dY);
            asm volatile("" : "+s"(b));
            st.sv_base[f] = (vf_gptr)b;
            st.sv_off[f] = ((unsigned)rc * (unsigned)E.ld_dy + 4u * gq) * 4u;                   // bytes
        }
    }
}

template <class P, int OI, int LOCAL>
__device__ __forceinline__ void bwd16_deferred_store(const BwdState16<P>& st)
{
    if constexpr (OI >= 1) {
        constexpr BwdOp Q = P::op(OI - 1);
        constexpr int S0 = Q.nfin > 0 ? Q.fin[0].nt * 2 : 0, S = S0 + (Q.nfin > 1 ? Q.fin[1].nt * 2 : 0);
        constexpr int n_it = Bwd16<P>::items(OI), per = (S + n_it - 1) / n_it;
        constexpr int s0 = LOCAL * per, s1 = (LOCAL + 1) * per < S ? (LOCAL + 1) * per : S;
        if constexpr (s0 < s1) {
#pragma unroll
            for (int i = s0; i < s1; ++i) {
                const int f = i < S0 ? 0 : 1, a = i - (f ? S0 : 0);
                const f32x4& v = st.t[2 * Q.fin[f].t0 + a];
                *(vf_gfloat4*)(st.sv_base[f] + st.sv_off[f] + 16 * a * 4) = vf_st4{v[0], v[1], v[2], v[3]};
            }
        }
    }
}

// (items (T, a), (T, a + 1) as one step with their MFMAs interleaved: see chain16_items)
template <class P, int I, bool PRE = false>
__device__ __forceinline__ void bwd16_items(const BwdArgsChain& g, BwdState16<P>& st, int lane, int row, int rc, bool live)
{
    using B = Bwd16<P>;
    if constexpr (I < B::n_items()) {
        constexpr int oi = B::op_of(I), local = I - B::first_item(oi);
        constexpr BwdOp O = P::op(oi);
        constexpr int T = local / B::nout(oi), a = local % B::nout(oi);
        constexpr bool pair = O.in_kind == 0 && (a % 2 == 0) && (a + 1 < B::nout(oi));
        constexpr int step = pair ? 2 : 1;
        const int gq = lane >> 4;
        const float4 w = st.ring[I % kChain16Depth];
        if constexpr (I + kChain16Depth < B::n_items()) st.ring[I % kChain16Depth] = bwd16_load<P, I + kChain16Depth>(g, lane);
        float4 w1 = w;
        if constexpr (pair) {
            w1 = st.ring[(I + 1) % kChain16Depth];
            if constexpr (I + 1 + kChain16Depth < B::n_items()) st.ring[(I + 1) % kChain16Depth] = bwd16_load<P, I + 1 + kChain16Depth>(g, lane);
        }
        if constexpr (local == 0 && O.in_kind == 0 && !PRE) bwd16_mask_load<P, oi>(g, st, rc, gq);    // head ops: in the prologue
        f32x4& acc = st.t[2 * O.out0 + a];
        f32x4& acc1 = st.t[2 * O.out0 + a + (pair ? 1 : 0)];
        if constexpr (T == 0 && !O.accum) {
            acc = f32x4{0};
            if constexpr (pair) acc1 = f32x4{0};
        }
        if constexpr (O.in_kind == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float b = st.t[2 * O.in0 + T][j];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(j == 0 ? w.x : j == 1 ? w.y : j == 2 ? w.z : w.w, b, acc, 0, 0, 0);
                if constexpr (pair) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(j == 0 ? w1.x : j == 1 ? w1.y : j == 2 ? w1.z : w1.w, b, acc1, 0, 0, 0);
            }
        } else {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, st.hin[O.in_kind - 1], acc, 0, 0, 0);
        }
        if constexpr (local == 0) bwd16_store_setup<P, oi>(g, st, rc, gq);
        bwd16_deferred_store<P, oi, local>(st);
        if constexpr (pair) bwd16_deferred_store<P, oi, local + 1>(st);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (local + step - 1 == B::items(oi) - 1) bwd16_finalize<P, oi>(g, st, row, gq, live);
        bwd16_items<P, I + step, PRE>(g, st, lane, row, rc, live);
    }
}

template <class P, int I>
__device__ __forceinline__ void bwd16_prologue(const BwdArgsChain& g, BwdState16<P>& st, int lane)
{
    if constexpr (I < kChain16Depth && I < Bwd16<P>::n_items()) {
        st.ring[I] = bwd16_load<P, I>(g, lane);
        bwd16_prologue<P, I + 1>(g, st, lane);
    }
}

template <class P, int OI, bool PRE = false>
__device__ __forceinline__ void bwd16_head_prologue(const BwdArgsChain& g, BwdState16<P>& st, int rc, int gq, bool live)
{
    if constexpr (OI < P::n_ops) {
        constexpr BwdOp O = P::op(OI);
        if constexpr (O.in_kind != 0) {
            const vf_mlp_bwd_layer& E = g.d.layer[P::entry(O.fl)];
            const float* dy = E.dY + (size_t)rc * E.ld_dy;
            if constexpr (O.in_kind == 1) {
                float4 dm;
                if (g.rp_d_action) {       // k_reparam_bwd's arithmetic; lane group 0 of a live row writes d_mean / g_log_std
                    const float4 da = PRE ? st.pda : g.rp_d_action[rc], a = PRE ? st.pa : g.rp_action[rc], e = PRE ? st.pe : g.rp_eps[rc];
                    dm = make_float4(da.x * (1.0f - a.x * a.x), da.y * (1.0f - a.y * a.y), da.z * (1.0f - a.z * a.z), da.w * (1.0f - a.w * a.w));
                    if constexpr (P::sac_head) {       // k_shac_head_bwd's: d_mu = dm, d_log_std (the second head's gradient) from the saved row
                        const vf_mlp_bwd_layer& EV = g.d.layer[P::entry(P::L_val)];
                        const float4 s = PRE ? st.pg : g.rp_ls_rows[rc];
                        const float lo = g.rp_ls_lo, hi = g.rp_ls_hi;
                        const float4 dl = make_float4(sac_dls(dm.x, e.x, s.x, lo, hi), sac_dls(dm.y, e.y, s.y, lo, hi),
                                                      sac_dls(dm.z, e.z, s.z, lo, hi), sac_dls(dm.w, e.w, s.w, lo, hi));
                        if (live && gq == 0) {
                            *reinterpret_cast<float4*>(const_cast<float*>(E.dY) + (size_t)rc * E.ld_dy) = dm;
                            *reinterpret_cast<float4*>(const_cast<float*>(EV.dY) + (size_t)rc * EV.ld_dy) = dl;
                        }
                        st.hin[1] = gq == 0 ? dl.x : gq == 1 ? dl.y : gq == 2 ? dl.z : dl.w;
                    } else if (live && gq == 0) {
                        *reinterpret_cast<float4*>(const_cast<float*>(E.dY) + (size_t)rc * E.ld_dy) = dm;
                        float4 gl = PRE ? st.pg : g.rp_g_log_std[rc];
                        float sd[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) sd[k] = g.rp_std_valid ? g.rp_std[k] : expf(g.rp_log_std[k]);
                        gl.x += dm.x * sd[0] * e.x; gl.y += dm.y * sd[1] * e.y;
                        gl.z += dm.z * sd[2] * e.z; gl.w += dm.w * sd[3] * e.w;
                        g.rp_g_log_std[rc] = gl;
                    }
                } else if constexpr (P::Net::HM == 1) {
                    dm = make_float4(dy[0], 0.0f, 0.0f, 0.0f);
                } else {
                    dm = make_float4(dy[0], dy[1], dy[2], dy[3]);
                }
                st.hin[0] = gq == 0 ? dm.x : gq == 1 ? dm.y : gq == 2 ? dm.z : dm.w;
            } else if constexpr (P::Net::HV == 4) {
                if (!(P::sac_head && g.rp_d_action)) st.hin[1] = dy[gq];      // (else: formed by the first head's branch above)
            } else {
                st.hin[1] = gq == 0 ? dy[0] : 0.0f;
            }
            if constexpr (!PRE) bwd16_mask_load<P, OI>(g, st, rc, gq);
            bwd16_head_prologue<P, OI + 1, PRE>(g, st, rc, gq, live);
        }
    }
}

// the tile that holds dLoss / d observation 0 (the "state" branch) after the sweep: lane (m, gq) = features 4 gq .. 4 gq + 3 of row m
template <class P>
__device__ __forceinline__ const f32x4& bwd16_obs_tile(const BwdState16<P>& st)
{
    constexpr int oi = [] { for (int i = 0; i < P::n_ops; ++i) if (P::op(i).obs == 0) return i; return -1; }();
    static_assert(oi >= 0, "the program has no observation-gradient op");
    return st.t[2 * P::op(oi).out0];
}

// the last op's finalised tiles have no following items to carry their stores
template <class P>
__device__ __forceinline__ void bwd16_tail_store(const BwdArgsChain& g, const BwdState16<P>& st, int row, int gq, bool live)
{
    constexpr BwdOp Q = P::op(P::n_ops - 1);
    if (!live) return;
#pragma unroll
    for (int f = 0; f < Q.nfin; ++f) {
        const vf_mlp_bwd_layer& E = g.d.layer[P::entry(Q.fin[f].fl)];
        float* base = const_cast<float*>(E.dY) + (size_t)row * E.ld_dy + 4 * gq;
#pragma unroll
        for (int a = 0; a < 2 * Q.fin[f].nt; ++a) {
            const f32x4& v = st.t[2 * Q.fin[f].t0 + a];
            *reinterpret_cast<float4*>(base + 16 * a) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// one reverse pass of the rows `row` (lane & (ROWS - 1) of the wave), ROWS = 32 or 16
template <class P, int ROWS>
__device__ __forceinline__ void bwd_rows(const BwdArgsChain& g, int lane, int row, int rc, bool live)
{
    if constexpr (ROWS == 32) {
        const int h = lane >> 5;
        BwdState<P> st;
        bwd_prologue<P, 0>(g, st, lane);
        bwd_head_prologue<P, 0>(g, st, rc, h, live);
        bwd_items<P, NoFwd, 0>(g, st, NoFwd{}, lane, row, rc, live);
        bwd_tail_store<P>(g, st, row, h, live);
    } else {
        const int gq = lane >> 4;
        BwdState16<P> st;
        bwd16_prologue<P, 0>(g, st, lane);
        bwd16_head_prologue<P, 0>(g, st, rc, gq, live);
        bwd16_items<P, 0>(g, st, lane, row, rc, live);
        bwd16_tail_store<P>(g, st, row, gq, live);
    }
}

}  // namespace vf
