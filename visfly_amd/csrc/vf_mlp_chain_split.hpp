// vf_mlp_chain_split.hpp -- branch-parallel register chains: TWO waves share one tile of 32 rows and each walks HALF of the network.
//
// The networks of the reference's policies are two pipelines that meet once (utils/policies/extractors.py:662-678: state (+) target
// extractors -> features; policies.py:18-49: features -> pi trunk -> action mean  ||  features -> vf trunk -> value; the twin critic of
// td_policies.py:82-143: features (+) action -> qf0 || qf1).  The one-wave chain (vf_mlp_chain.hip) walks all of it serially: 1 408
// dependent MFMAs per PPO minibatch tile in 476 VGPRs, so one wave per SIMD and an MFMA pipe that idles whenever that wave does
// anything else (epilogues, stores, the loss).  Here a workgroup of two waves owns the tile:
//     role 0: [extractor 0 | first half of the only extractor] -> pi trunk -> first head (+ its part of the loss) -> reverse of the same
//     role 1: [extractor 1 | second half]                       -> vf trunk -> second head (+ its part)           -> reverse of the same
// Both waves use the SAME lane -> row mapping and accumulator layout, so what they owe each other is a lane-to-same-lane copy
// through LDS, once per direction: the feature tiles (forward) and the partial feature gradients (reverse, added).  Each wave holds
// about half of the tiles (<= 256 VGPRs), so two of them share a SIMD and fill each other's idle pipe.
//
//   NB = 2 (StateTargetExtractor): role r runs extractor branch r whole and exchanges its E2 feature tiles.
//   NB = 1 (StateExtractor): both roles run the cheap observation layer (K <= 16: 32 MFMAs), role r computes output tile(s) r of the
//           extractor's second layer and, in reverse, the half [r E1/2, (r + 1) E1/2) of that layer's input gradient.  Saved copies / dZ
//           rows of a layer computed by both are stored by the half that owns the tiles (ChainLayer::s0 / sn, BwdFin::s0 / sn).
//
// Bits: a forward value is the same sum of the same products in the same order as in the one-wave chain.  The feature gradient is
// (pi part) + (vf part) of two separately accumulated sums instead of one accumulator running through both: equal to rounding, like
// every other choice of kernel for the same rows (tests hold it to the bounds of the one-wave fused kernel).
#pragma once
#include "vf_mlp_chain_bwd.hpp"

namespace vf {

// the forward program of role R over ChainNet N (same tile numbering as N: a role simply never touches the other half's tiles)
template <class N, int R>
struct SplitNet : N {
    static_assert(N::VF, "both trunks");
    static_assert(N::NB == 2 || (N::E1 % 2 == 0 && N::E2 % 2 == 0), "one extractor: its layers are split by output tiles");
    static constexpr int role = R;
    static constexpr int NB = N::NB;
    static constexpr int n_exec = 5;
    // the first extractor layer's output is the longest-lived ReLU mask (from the head of the forward to the tail of the reverse): kept as
    // bits (ChainLayer::pk0) it costs 2 registers instead of 64, which is what puts the StateTarget class under 256 VGPRs without spills
    static constexpr bool kPackE1 = true;
    static constexpr int mask_bits(int fl) { return kPackE1 && fl == (NB == 2 ? 2 * R : 0) ? 0 : -1; }
    static constexpr int t_tr1 = R ? N::t_v1 : N::t_p1, t_tr2 = R ? N::t_v2 : N::t_p2, t_head = R ? N::t_val : N::t_mean;
    static constexpr int W1 = R ? N::V1 : N::P1, W2 = R ? N::V2 : N::P2;
    static constexpr ChainLayer layer(int i)
    {
        if (i == 0) {
            if (NB == 2) return ChainLayer{2 * R, R, 0, N::kin(R) / 8, N::t_e1(R), N::E1, 1};
            return ChainLayer{0, 0, 0, N::kin(0) / 8, N::t_e1(0), N::E1, 1, 0, R * (N::E1 / 2), N::E1 / 2};          // computed by both, stored by halves
        }
        if (i == 1) {
            if (NB == 2) return ChainLayer{2 * R + 1, -1, N::t_e1(R), N::E1, N::t_feat + R * N::E2, N::E2, 1, 0, 0, -1, N::E2, N::t_feat + (1 - R) * N::E2, kPackE1 ? 0 : -1};
            return ChainLayer{1, -1, N::t_e1(0), N::E1, N::t_feat + R * (N::E2 / 2), N::E2 / 2, 1, R * (N::E2 / 2), 0, -1, N::E2 / 2,
                              N::t_feat + (1 - R) * (N::E2 / 2), kPackE1 ? 0 : -1};
        }
        switch (i) {
        case 2: return ChainLayer{N::base + 3 * R, -1, N::t_feat, N::n_feat, t_tr1, W1, 1};
        case 3: return ChainLayer{N::base + 3 * R + 1, -1, t_tr1, W1, t_tr2, W2, 1};
        default: return ChainLayer{N::base + 3 * R + 2, -1, t_tr2, W2, t_head, 1, 0};
        }
    }
    // (the item bookkeeping of ChainNet, over THIS layer list)
    static constexpr int groups(int i) { return layer(i).obs >= 0 ? layer(i).nin : layer(i).nin * 4; }
    static constexpr int items(int i) { return groups(i) * layer(i).nout; }
    static constexpr int n_items()
    {
        int n = 0;
        for (int i = 0; i < n_exec; ++i) n += items(i);
        return n;
    }
    static constexpr bool is_head(int i) { return layer(i).desc == N::L_mean || layer(i).desc == N::L_value; }
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

// the reverse program of role R (both trunks, no observation gradient: the update kernels' variant)
template <class N, int R>
struct SplitBwd : BwdProg<N, true, true, false> {
    using B = BwdProg<N, true, true, false>;
    using Net = N;
    static constexpr int role = R;
    static constexpr int NB = N::NB;
    static constexpr int n_ops = 4;
    static constexpr int g_tr2 = R ? B::g_v2 : B::g_p2, g_tr1 = R ? B::g_v1 : B::g_p1;
    static constexpr int W1 = R ? N::V1 : N::P1, W2 = R ? N::V2 : N::P2;
    static constexpr int L_tr0 = R ? B::L_vf0 : B::L_pi0, L_tr1 = R ? B::L_vf1 : B::L_pi1, L_head = R ? B::L_val : B::L_mean;
    static constexpr BwdOp op(int i)
    {
        switch (i) {
        case 0: return BwdOp{L_head, 1 + R, 0, 1, g_tr2, W2, 0, -1, 1, {BwdFin{L_tr1, g_tr2, W2, 0}, {}}};
        case 1: return BwdOp{L_tr1, 0, g_tr2, W2 * 4, g_tr1, W1, 0, -1, 1, {BwdFin{L_tr0, g_tr1, W1, 0}, {}}};
        case 2:     // this trunk's part of the feature gradient; the partner's part arrives through LDS before the mask
            if (NB == 2)
                return BwdOp{L_tr0, 0, g_tr1, W1 * 4, B::g_feat, NB * N::E2, 0, -1, 1, {BwdFin{2 * R + 1, B::g_feat + R * N::E2, N::E2, 0}, {}},
                             0, N::E2, B::g_feat + (1 - R) * N::E2, B::g_feat + R * N::E2};
            return BwdOp{L_tr0, 0, g_tr1, W1 * 4, B::g_feat, N::E2, 0, -1, 1, {BwdFin{1, B::g_feat, N::E2, 0, 0, R * (N::E2 / 2), N::E2 / 2}, {}},
                         0, N::E2, B::g_feat, B::g_feat};
        default:
            if (NB == 2)
                return BwdOp{2 * R + 1, 0, B::g_feat + R * N::E2, N::E2 * 4, B::g_e1(R), N::E1, 0, -1, 1, {BwdFin{2 * R, B::g_e1(R), N::E1, 0}, {}}};
            return BwdOp{1, 0, B::g_feat, N::E2 * 4, B::g_e1(0) + R * (N::E1 / 2), N::E1 / 2, 0, -1, 1,
                         {BwdFin{0, B::g_e1(0) + R * (N::E1 / 2), N::E1 / 2, 0, R * (N::E1 / 2)}, {}}, R * (N::E1 / 2)};
        }
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

// observation fragments of the branches role R reads (NB = 2: its own branch only)
template <class N, int R>
// rs: the buffer row the lane loads; rc_copy / live: where (and whether) the row is also written in call order (ChainArgs::obs_copy: the
// fused PPO step on an indexed minibatch, vf_ppo_loss_cfg.row_index).  One extractor: both roles load the row, role 0 writes the copy
__device__ __forceinline__ void split_load_obs(const ChainArgs& g, ChainState<SplitNet<N, R>>& fs, int rs, int h, int rc_copy = 0, bool live = false)
{
#pragma unroll
    for (int b = 0; b < N::NB; ++b) {
        if (N::NB == 2 && b != R) continue;
        const int w = g.d.in_dim[b];
        const float* x = g.io.in[b] + (size_t)rs * w;
        float* xc = (g.obs_copy[b] && live && (N::NB == 2 || R == 0)) ? g.obs_copy[b] + (size_t)rc_copy * w : nullptr;
#pragma unroll
        for (int s = 0; s < N::kin(b) / 2; ++s) {
            const int k = 2 * s + h;
            const float v = x[k < w ? k : w - 1];
            fs.x[b][s] = k < w ? v : 0.0f;
            if (xc && k < w) xc[k] = v;
        }
    }
}

}  // namespace vf
