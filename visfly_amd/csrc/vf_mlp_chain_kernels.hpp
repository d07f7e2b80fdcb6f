// vf_mlp_chain_kernels.hpp -- the register-chained forward / reverse kernels as templates over the network class, with their layer-table
// matchers and launchers, shared by the translation units that instantiate them: vf_mlp_chain.hip (the actor-critic classes of the
// reference's PPO policies and the policy-only classes) and vf_mlp_chain_sac.hip (the SAC-style Actor of utils/policies/td_policies.py).
// Two files only so that the ~2 minutes of fully unrolled code per class compile in parallel.  Scheme: see vf_mlp_chain.hip.
#pragma once
#include "vf_mlp_chain_bwd.hpp"

namespace vf {

struct PpoRowArgs {          // per-row inputs of the fused PPO minibatch step (k_ppo_update_chain / k_ppo_update_split)
    const float* log_std;
    const float4* action;
    const float* old_lp;
    const float* adv;
    const float* ret;
    float* part;             // [n_tiles][kStats]
    vf_ppo_loss_cfg cfg;     // cfg.row_index: the call's row m is row row_index[m] of the observation / action / old_lp / ret / old_value buffers
};

// the buffer row a lane's row comes from (vf_ppo_loss_cfg.row_index; one dependent load at the head of the kernel)
__device__ __forceinline__ int ppo_source_row(const PpoRowArgs& pr, int rc) { return pr.cfg.row_index ? (int)pr.cfg.row_index[rc] : rc; }

// vf_mlp_chain_split.hip: the fused update kernels with two waves per row tile; 1 launched, 0 not taken, < 0 error
int ppo_update_split_try(const ChainArgs& g, const BwdArgsChain& gb, const void* pr, int which, int M, hipStream_t st);
int twin_q_update_split_try(const ChainArgs& g, const BwdArgsChain& gb, const float* target, double* part, float scale, int M, hipStream_t st);

template <class N>
__global__ __launch_bounds__(64) void k_mlp_forward_chain(const ChainArgs g)
{
    prefetch_kernarg<sizeof(ChainArgs)>();
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
        float* xc = g.obs_copy[b] ? g.obs_copy[b] + (size_t)rc * w : nullptr;
#pragma unroll
        for (int s = 0; s < N::kin(b) / 2; ++s) {
            const int k = 2 * s + h;
            const float v = x[k < w ? k : w - 1];
            st.x[b][s] = k < w ? v : 0.0f;
            if (xc && live && k < w) xc[k] = v;
        }
    }
    chain_pass_tile<N>(g, st, row, rc, h, live);
    chain_items<N, 0>(g, st, lane, row, live, rc);
}

template <class N>
__global__ __launch_bounds__(64) void k_mlp_forward_chain16(const ChainArgs g)
{
    VF_TRACE(0);
    prefetch_kernarg<sizeof(ChainArgs)>();
    const int lane = threadIdx.x, m = lane & 15, gq = lane >> 4;
    const int row = blockIdx.x * 16 + m;
    const bool live = row < g.M;
    const int rc = live ? row : g.M - 1;
    ChainState16<N> st;
    chain16_prologue<N, 0>(g, st, lane);
#pragma unroll
    for (int b = 0; b < N::NB; ++b) {
        const int w = g.d.in_dim[b];
        const float* x = g.io.in[b] + (size_t)rc * w;
        float* xc = g.obs_copy[b] ? g.obs_copy[b] + (size_t)rc * w : nullptr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * gq + j;
            const float v = x[k < w ? k : w - 1];
            st.x[b][j] = k < w ? v : 0.0f;
            if (xc && live && k < w) xc[k] = v;
        }
    }
    chain16_pass_tile<N>(g, st, row, rc, gq, live);
    VF_TRACE(1);
    chain16_items<N, 0>(g, st, lane, row, live, rc);
    VF_TRACE(31);
}

template <class P, int ROWS = 32>
__global__ __launch_bounds__(64) void k_mlp_backward_chain(const BwdArgsChain g)
{
    prefetch_kernarg<sizeof(BwdArgsChain)>();
    const int lane = threadIdx.x, m = lane & (ROWS - 1);
    const int row = blockIdx.x * ROWS + m;
    const bool live = row < g.M;
    const int rc = live ? row : g.M - 1;
    bwd_rows<P, ROWS>(g, lane, row, rc, live);
}

// ------------------------------------------------------------------------------------------------
// PPO minibatch step, everything that is per-row in ONE launch: forward chain -> clipped-surrogate loss of the wave's
// 32 rows (their head outputs never leave the registers) -> reverse chain, whose ReLU masks are the forward's own
// accumulator tiles (still live), so nothing is read back.  Left in HBM for the weight-gradient kernel: the layer
// inputs X (forward's saved copies), the masked gradients dZ, d_mean / d_value; per wave one row of loss-statistic
// partials (folded by the loss kernel's k_fold_stats).
// ------------------------------------------------------------------------------------------------
template <class N>
__global__ __launch_bounds__(64) void k_ppo_update_chain(const ChainArgs g, const BwdArgsChain gb, const PpoRowArgs pr)
{
    using P = typename N::template Bwd<true, true, false>;
    prefetch_kernarg<sizeof(ChainArgs) + sizeof(BwdArgsChain) + sizeof(PpoRowArgs)>();
    const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
    const int row = blockIdx.x * 32 + m;
    const bool live = row < g.M;
    const int rc = live ? row : g.M - 1;
    // per-row loss inputs first: the action-only part of the loss (ppo_row_pre) runs while the weight fragments are on their way
    const int rs = ppo_source_row(pr, rc);
    const float4 a4 = pr.action[rs];
    const float ls[4] = {pr.log_std[0], pr.log_std[1], pr.log_std[2], pr.log_std[3]};
    const float old_lp = pr.old_lp[rs], adv = pr.adv[rc], ret = pr.ret[rs];
    ChainState<N> fs;
    if constexpr (N::pack_or) {
#pragma unroll
        for (int i = 0; i < N::n_mb; ++i) fs.mb[i] = 0u;
    }
    chain_prologue<N, 0>(g, fs, lane);
#pragma unroll
    for (int b = 0; b < N::NB; ++b) {
        const int w = g.d.in_dim[b];
        const float* x = g.io.in[b] + (size_t)rs * w;
        float* xc = g.obs_copy[b] ? g.obs_copy[b] + (size_t)rc * w : nullptr;      // (row_index: the rows in call order, for the weight gradients)
#pragma unroll
        for (int s = 0; s < N::kin(b) / 2; ++s) {
            const int k = 2 * s + h;
            const float v = x[k < w ? k : w - 1];
            fs.x[b][s] = k < w ? v : 0.0f;
            if (xc && live && k < w) xc[k] = v;
        }
    }
    const float a[4] = {a4.x, a4.y, a4.z, a4.w};
    PpoRowPre pre = ppo_row_pre(a);
#pragma unroll
    for (int d = 0; d < 4; ++d)     // pinned here: left alone the compiler sinks the arithmetic to its use behind the forward chain
        asm volatile("" : "+v"(pre.g[d]), "+v"(pre.corr[d]));
    chain_items<N, 0>(g, fs, lane, row, live, rc);
    BwdState<P> bs;
    bwd_prologue<P, 0>(gb, bs, lane);                  // first weight blocks of the reverse chain: in flight during the loss arithmetic
    // ---- loss of this lane's row (lane half 0 holds mean[0..3] / value in registers 0..3 / 0 of the head tiles) ----
    float stt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dm[4] = {0, 0, 0, 0}, dvl = 0.0f;
    {
        const f32x16& mt = fs.t[N::t_mean];
        const float mu[4] = {mt[0], mt[1], mt[2], mt[3]};
        float st1[9], dm1[4], dv1;
        ppo_row_post(pre, mu, fs.t[N::t_val][0], ls, old_lp, adv, ret, pr.cfg, dm1, dv1, st1, rs);
        const bool on = live && h == 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) stt[k] = on ? st1[k] : 0.0f;
        // head gradients: lane half 0 of EVERY lane -- the lanes past the last row are replicas of row M - 1 (they loaded its inputs)
        // and must stay replicas through the reverse chain, whose dZ stores are unguarded (bwd_store_setup): they rewrite that row's
        // values, they do not zero them.  Only the statistics above and the head rows below exclude them.
#pragma unroll
        for (int k = 0; k < 4; ++k) dm[k] = h == 0 ? dm1[k] : 0.0f;
        dvl = h == 0 ? dv1 : 0.0f;
        if (on) {         // head gradients: dZ of the head layers for the weight-gradient kernel
            const vf_mlp_bwd_layer& Em = gb.d.layer[P::entry(P::L_mean)];
            const vf_mlp_bwd_layer& Ev = gb.d.layer[P::entry(P::L_val)];
            *reinterpret_cast<float4*>(const_cast<float*>(Em.dY) + (size_t)row * Em.ld_dy) = make_float4(dm[0], dm[1], dm[2], dm[3]);
            const_cast<float*>(Ev.dY)[(size_t)row * Ev.ld_dy] = dvl;
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        float s = stt[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) pr.part[(size_t)blockIdx.x * kStats + k] = s;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) bs.hin[0][k] = dm[k];
    bs.hin[1][0] = dvl; bs.hin[1][1] = 0.0f; bs.hin[1][2] = 0.0f; bs.hin[1][3] = 0.0f;
    bwd_items<P, ChainState<N>, 0>(gb, bs, fs, lane, row, rc, live);
    bwd_tail_store<P>(gb, bs, row, h, live);
}

// ------------------------------------------------------------------------------------------------
// One critic update of SHAC (shac.py:267-270), everything that is per-row in ONE launch: forward chain of the twin critic ->
// mse_loss(returns, min(Q1, Q2)) of the wave's 32 rows and its gradient w.r.t. the two heads (k_twin_q_loss's arithmetic; the head
// outputs never leave the registers) -> reverse chain, whose ReLU masks are the forward's own accumulator tiles (still live), so the
// saved activations are not read back: k_ppo_update_chain's scheme for the critic class.  Left in HBM for the weight-gradient launch:
// the layer inputs X, the masked gradients dZ, dQ1 / dQ2; per wave one fp64 partial of the squared error (k_twin_q_fold).
// ------------------------------------------------------------------------------------------------
template <class N>
__global__ __launch_bounds__(64) void k_twin_q_update_chain(const ChainArgs g, const BwdArgsChain gb, const float* __restrict__ target,
                                                            double* __restrict__ part, float scale)
{
    using P = typename N::template Bwd<true, true, false>;
    prefetch_kernarg<sizeof(ChainArgs) + sizeof(BwdArgsChain) + 24>();
    const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
    const int row = blockIdx.x * 32 + m;
    const bool live = row < g.M;
    const int rc = live ? row : g.M - 1;
    ChainState<N> fs;
    if constexpr (N::pack_or) {
#pragma unroll
        for (int i = 0; i < N::n_mb; ++i) fs.mb[i] = 0u;
    }
    chain_prologue<N, 0>(g, fs, lane);
#pragma unroll
    for (int b = 0; b < N::NB; ++b) {
        const int w = g.d.in_dim[b];
        const float* x = g.io.in[b] + (size_t)rc * w;
#pragma unroll
        for (int s = 0; s < N::kin(b) / 2; ++s) {
            const int k = 2 * s + h;
            const float v = x[k < w ? k : w - 1];
            fs.x[b][s] = k < w ? v : 0.0f;
        }
    }
    chain_pass_tile<N>(g, fs, row, rc, h, live);
    const float tgt = target[rc];                       // issued before the forward so that it has arrived when the forward ends
    chain_items<N, 0>(g, fs, lane, row, live, rc);
    BwdState<P> bs;
    bwd_prologue<P, 0>(gb, bs, lane);                  // first weight blocks of the reverse chain: in flight during the loss arithmetic
    // ---- loss of this lane's row (lane half 0 holds Q1 / Q2 in register 0 of the head tiles) ----
    const float q0 = fs.t[N::t_mean][0], q1 = fs.t[N::t_val][0];
    const bool first = q0 <= q1;                       // ties: the first, like torch.min over dim 1
    const float diff = (first ? q0 : q1) - tgt;
    const float gq = 2.0f * diff * scale;
    // head gradients: lane half 0 of EVERY lane -- lanes past the last row are replicas of row M - 1 and stay replicas through the
    // reverse chain, whose dZ stores are unguarded (k_ppo_update_chain); only the partial sum and the head rows exclude them
    const float dq0 = h == 0 ? (first ? gq : 0.0f) : 0.0f, dq1 = h == 0 ? (first ? 0.0f : gq) : 0.0f;
    double sq = (live && h == 0) ? (double)diff * (double)diff : 0.0;
    if (live && h == 0) {                               // dZ of the head layers for the weight-gradient kernel
        const vf_mlp_bwd_layer& E0 = gb.d.layer[P::entry(P::L_mean)];
        const vf_mlp_bwd_layer& E1 = gb.d.layer[P::entry(P::L_val)];
        const_cast<float*>(E0.dY)[(size_t)row * E0.ld_dy] = dq0;
        const_cast<float*>(E1.dY)[(size_t)row * E1.ld_dy] = dq1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
    if (lane == 0) part[blockIdx.x] = sq;
    bs.hin[0][0] = dq0; bs.hin[0][1] = 0.0f; bs.hin[0][2] = 0.0f; bs.hin[0][3] = 0.0f;
    bs.hin[1][0] = dq1; bs.hin[1][1] = 0.0f; bs.hin[1][2] = 0.0f; bs.hin[1][3] = 0.0f;
    bwd_items<P, ChainState<N>, 0>(gb, bs, fs, lane, row, rc, live);
    bwd_tail_store<P>(gb, bs, row, h, live);
}

// does the layer table describe network class N (shapes, wiring, execution order of MlpPolicy)?
template <class N>
bool chain_matches(const vf_mlp_desc& d)
{
    if (d.n_layers != N::n_layers || d.n_inputs != N::NB + N::PASS) return false;
    for (int b = 0; b < N::NB; ++b)
        if (d.in_dim[b] < 1 || d.in_dim[b] > N::kin(b)) return false;
    auto is = [&](int li, int K, int No, int relu) {
        const vf_mlp_layer& L = d.layer[li];
        return L.K == K && L.No == No && L.relu == (relu ? VF_ACTIVATION_RELU : VF_ACTIVATION_NONE) && L.wr_off >= 0 && (L.wr_off & 3) == 0;     // (the built-in classes are ReLU networks)
    };
    int feat = N::NB * N::E2 * 32;
    for (int b = 0; b < N::NB; ++b) {
        if (!is(2 * b, d.in_dim[b], N::E1 * 32, 1) || !is(2 * b + 1, N::E1 * 32, N::E2 * 32, 1)) return false;
        const vf_mlp_layer &l1 = d.layer[2 * b], &l2 = d.layer[2 * b + 1];
        if (l1.src != b || l1.src_col != 0 || l2.src != l1.dst || l2.src_col != l1.dst_col) return false;
        if (l2.dst_col != b * N::E2 * 32 || l2.dst != d.layer[1].dst) return false;
    }
    const int fid = d.layer[1].dst;
    if constexpr (N::PASS) {      // the frozen identity layer: input NB, <= 4 columns, appended to the features, declared in identity_mask
        const vf_mlp_layer& I = d.layer[2 * N::NB];
        const int pw = d.in_dim[N::NB];
        if (pw < 1 || pw > 4 || I.K != pw || I.No != pw || I.relu || I.src != N::NB || I.src_col != 0 || I.dst != fid || I.dst_col != feat) return false;
        if (!((d.identity_mask >> (2 * N::NB)) & 1)) return false;
        if (I.save && ((I.save_ld & 3) || (I.dst_col & 3))) return false;
        feat += pw;
    } else if (d.identity_mask) {
        return false;
    }
    const int w1[2] = {N::P1 * 32, N::V1 * 32}, w2[2] = {N::P2 * 32, N::V2 * 32}, wo[2] = {N::HM, N::HV};
    for (int t = 0; t < 2; ++t) {
        const int l = N::base + 3 * t;
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

// 16 rows per wave: only while it doubles the waves without exceeding one per SIMD (M <= 16 384), K <= 16 observation rows,
// and the weight rows the kernel reads as float4 are 16-byte aligned.  VISFLY_AMD_MLP_CHAIN16=0/1 forces the choice (A/B).
template <class N>
bool chain16_ok(const vf_mlp_desc& d, const float* params, int M)
{
    static const int forced = [] { const char* e = getenv("VISFLY_AMD_MLP_CHAIN16"); return e ? atoi(e) : -1; }();
    if (forced == 0 || (forced < 0 && M > 16384)) return false;
    if (reinterpret_cast<uintptr_t>(params) & 15) return false;
    for (int b = 0; b < N::NB; ++b)
        if (d.in_dim[b] > 16) return false;
    for (int i = 0; i < N::n_exec; ++i) {
        const vf_mlp_layer& L = d.layer[N::layer(i).desc];
        if (N::layer(i).obs < 0 && ((L.w_off & 3) || (L.K & 15))) return false;
        if (L.wt_off < 0) return false;               // the transposed image the A fragments come from (chain16_load)
    }
    return true;
}

// M_choice > 0: the rows-per-wave choice is made for M_choice rows (vf_mlp_forward_steps: n consecutive blocks of M_choice rows in one
// launch, each row computed exactly as a launch over its block alone would)
template <class N>
int chain_launch(const vf_mlp_desc& d, const float* params, const float* packed, const float* in0, const float* in1, float* out0,
                 float* out1, int M, hipStream_t st, const ReparamFwd& rp, const float* in2 = nullptr, int M_choice = 0)
{
    ChainArgs g{d, params, packed, ChainIo{{in0, in1, in2}, out0, out1}, M, rp.log_std, reinterpret_cast<const float4*>(rp.eps), reinterpret_cast<float4*>(rp.action),
                {rp.obs_copy[0], rp.obs_copy[1]}};
    if (chain16_ok<N>(d, params, M_choice > 0 ? M_choice : M))
        hipLaunchKernelGGL(k_mlp_forward_chain16<N>, dim3((M + 15) / 16), dim3(64), 0, st, g);
    else
        hipLaunchKernelGGL(k_mlp_forward_chain<N>, dim3((M + 31) / 32), dim3(64), 0, st, g);
    VF_HIP(hipGetLastError());
    return 1;
}

template <class N, bool PI, bool VF, bool IG>
bool bwd_chain_matches(const vf_mlp_bwd_desc& d)
{
    using P = BwdProg<N, PI, VF, IG>;
    if (d.n_layers != 2 * N::NB + 3 * ((PI ? 1 : 0) + (VF ? 1 : 0))) return false;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    auto is = [&](int fl, int K, int No, bool relu, bool first) {
        const vf_mlp_bwd_layer& E = d.layer[P::entry(fl)];
        if (E.K != K || E.No != No || (E.Y != nullptr) != relu || E.act > VF_ACTIVATION_RELU || E.wq_off < 0 || (E.wq_off & 3)) return false;
        if (relu && (!al16(E.Y) || (E.ld_y & 3) || !al16(E.dY) || (E.ld_dy & 3))) return false;   // float4 mask loads / dZ stores
        if (first ? (E.need_dx != 0) != IG : E.need_dx == 0) return false;
        return true;
    };
    for (int b = 0; b < N::NB; ++b) {
        const int K0 = d.layer[P::entry(2 * b)].K;
        if (K0 < 1 || K0 > N::kin(b) || K0 > 32) return false;
        if (!is(2 * b, K0, N::E1 * 32, true, true) || !is(2 * b + 1, N::E1 * 32, N::E2 * 32, true, false)) return false;
    }
    int feat = N::NB * N::E2 * 32;
    if constexpr (N::PASS) {       // features (+) pass-through columns: the trunks' first layers are K = feat + pw wide, pw = 1 .. 4
        const int K0 = d.layer[P::entry(PI ? P::L_pi0 : P::L_vf0)].K;
        if (K0 <= feat || K0 > feat + 4) return false;
        feat = K0;
    }
    if (PI && (!is(P::L_pi0, feat, N::P1 * 32, true, false) || !is(P::L_pi1, N::P1 * 32, N::P2 * 32, true, false) ||
               !is(P::L_mean, N::P2 * 32, N::HM, false, false)))
        return false;
    if (VF && (!is(P::L_vf0, feat, N::V1 * 32, true, false) || !is(P::L_vf1, N::V1 * 32, N::V2 * 32, true, false) ||
               !is(P::L_val, N::V2 * 32, N::HV, false, false)))
        return false;
    // wiring: the gradient a layer's consumer produces is that layer's dY buffer (incl. the feature concat)
    for (int b = 0; b < N::NB; ++b) {
        if (d.layer[P::entry(2 * b + 1)].dX != d.layer[P::entry(2 * b)].dY) return false;
        const vf_mlp_bwd_layer& first_trunk = d.layer[P::entry(PI ? P::L_pi0 : P::L_vf0)];
        if (d.layer[P::entry(2 * b + 1)].dY != first_trunk.dX + b * N::E2 * 32 || d.layer[P::entry(2 * b + 1)].ld_dy != first_trunk.ld_dx) return false;
    }
    if (PI && (d.layer[P::entry(P::L_pi1)].dX != d.layer[P::entry(P::L_pi0)].dY || d.layer[P::entry(P::L_mean)].dX != d.layer[P::entry(P::L_pi1)].dY))
        return false;
    if (VF && (d.layer[P::entry(P::L_vf1)].dX != d.layer[P::entry(P::L_vf0)].dY || d.layer[P::entry(P::L_val)].dX != d.layer[P::entry(P::L_vf1)].dY))
        return false;
    if (PI && VF && d.layer[P::entry(P::L_pi0)].dX != d.layer[P::entry(P::L_vf0)].dX) return false;
    return true;
}

// 16 rows per wave for the reverse chain?  The forward's rule (chain16_ok): a small row count leaves half of the SIMDs without a
// wave; only the policy-trunk variant with observation gradient (first-order policy optimisation, whose shards are small) is
// instantiated.  Needs the row-major data-gradient image (wb_off) and observation widths <= 16.  VISFLY_AMD_MLP_CHAIN16=0/1
// forces the choice (A/B) together with the forward's.
template <class N, bool PI, bool VF, bool IG>
bool bwd16_ok(const vf_mlp_bwd_desc& d, int M)
{
    using P = BwdProg<N, PI, VF, IG>;
    if constexpr (!(IG && ((PI && !VF) || P::sac_head))) return false;      // the classes a BPTT sweep runs per step (observation gradient)
    static const int forced = [] { const char* e = getenv("VISFLY_AMD_MLP_CHAIN16"); return e ? atoi(e) : -1; }();
    if (forced == 0 || (forced < 0 && M > 16384)) return false;
    for (int l = 0; l < d.n_layers; ++l)
        if (d.layer[l].wb_off < 0) return false;
    for (int b = 0; b < N::NB; ++b)
        if (d.layer[P::entry(2 * b)].K > 16) return false;
    return true;
}

template <class N, bool PI, bool VF, bool IG>
int bwd_chain_launch(const vf_mlp_bwd_desc& d, const float* packed, int M, hipStream_t st, const ReparamBwd& rp)
{
    BwdArgsChain g{d, packed, M, reinterpret_cast<const float4*>(rp.d_action), reinterpret_cast<const float4*>(rp.action), rp.log_std,
                   reinterpret_cast<const float4*>(rp.eps), reinterpret_cast<float4*>(rp.g_log_std)};
    if constexpr (IG && ((PI && !VF) || BwdProg<N, PI, VF, IG>::sac_head)) {
        if (bwd16_ok<N, PI, VF, IG>(d, M)) {
            hipLaunchKernelGGL((k_mlp_backward_chain<BwdProg<N, PI, VF, IG>, 16>), dim3((M + 15) / 16), dim3(64), 0, st, g);
            VF_HIP(hipGetLastError());
            return 1;
        }
    }
    hipLaunchKernelGGL((k_mlp_backward_chain<BwdProg<N, PI, VF, IG>>), dim3((M + 31) / 32), dim3(64), 0, st, g);
    VF_HIP(hipGetLastError());
    return 1;
}

// data gradients of the whole network (masked dZ of every hidden layer left in the dY buffers, optional observation
// gradients): 1 launched, 0 not an instantiated class / variant, < 0 error.  Variants: PPO update (both trunks, no
// observation gradient) and first-order policy optimisation (policy trunk only, observation gradient).
// the chains address their activation / gradient stores with 32-bit BYTE offsets (row * ld * 4): rows x widest row < 4 GiB
static inline bool rows_fit_u32(int M, int ld_max) { return (unsigned long long)M * (unsigned long long)ld_max * 4ull < (1ull << 32); }

}  // namespace vf
