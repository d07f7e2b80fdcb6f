// vf_mlp_chain_gen.hpp -- the register-chained kernels for ANY layer list of the reference's actor-critic policies
// (utils/policies/extractors.py:376-449 `create_mlp` / `set_mlp_feature_extractor`: a ReLU MLP of any depth per observation key;
// SB3 MlpExtractor: `net_arch=dict(pi=[...], vf=[...])` of any depth).  The classes instantiated in libvisfly_amd.so (NetHover, NetNav, ...)
// are the YAML-default shapes; every other shape gets its instance compiled on first use (visfly_amd/_jit.py: hipcc on a generated
// translation unit that names a Spec, cached next to the library, loaded through vf_chain_plugin_load).
//
// A Spec is a struct of compile-time constants:
//     NB            observation branches (1 or 2)
//     KIN[b]        first-layer input width of branch b padded to 8 (<= 32)
//     DE[b], EW[b][l]   layers of branch b's extractor and their widths in 32-feature tiles (<= 4 tiles = 128 features)
//     DP, PW[j]     hidden layers of the policy trunk (>= 1), widths in tiles;   DV, VW[j]: the value trunk's
//     VF            false: the class runs extractors + policy trunk only
//     HM, HV        widths of the two heads: (4, 1) the PPO policies' action mean / value; (4, 4) the SAC-style Actor's mu / log_std
//                   (utils/policies/td_policies.py:146-252: `pi` = latent_pi, `vf` = log_latent_pi); (1, 1) its twin ContinuousCritic's
//                   Q1 / Q2 (:82-143: `pi` = qf0, `vf` = qf1)
//     PASS          1: one more input of <= 4 columns (the critic's action) is appended to the features unchanged -- th.cat([features,
//                   actions]) (:137): one more input tile of the trunks' first layers (ChainNet::PASS); the layer table's frozen identity
//                   layer between the extractor layers and the trunks is not executed.  Layer numbers `fl` below are MlpPolicy order
//                   WITHOUT that layer (the numbering of the reverse chain); desc(fl) is the layer's index in the vf_mlp_desc
// ChainNetG<Spec> / BwdProgG<..> derive from it the same layer / op tables that ChainNet / BwdProg spell out for the two-layer shapes;
// the device code (vf_mlp_chain.hpp, vf_mlp_chain_bwd.hpp) is the same.
#pragma once
#include "vf_mlp_chain_kernels.hpp"

namespace vf {

constexpr int kGenMaxDepth = 4;      // layers per extractor branch / hidden layers per trunk
constexpr int kGenMaxOps = 4 * kGenMaxDepth + 4;

template <class S>
struct GenShape {
    static constexpr int NB = S::NB, DP = S::DP, DV = S::DV, PASS = S::PASS;
    static constexpr int de(int b) { return S::DE[b]; }
    static constexpr int ew(int b, int l) { return S::EW[b][l]; }
    // forward layers in MlpPolicy order: extractor branches layer by layer, policy trunk + mean head, value trunk + value head
    static constexpr int ext(int b, int l)
    {
        int n = 0;
        for (int i = 0; i < b; ++i) n += de(i);
        return n + l;
    }
    static constexpr int base() { return ext(NB, 0); }
    static constexpr int L_pi(int j) { return base() + j; }          // j == DP: the mean head
    static constexpr int L_mean() { return base() + DP; }
    static constexpr int L_vf(int j) { return L_mean() + 1 + j; }    // j == DV: the value head
    static constexpr int L_value() { return L_mean() + 1 + DV; }
    static constexpr int n_layers() { return L_value() + 1; }
    static constexpr int desc(int fl) { return fl + (PASS && fl >= base() ? 1 : 0); }       // index in the layer table (the identity layer sits at base())
    static constexpr bool is_head(int fl) { return fl == L_mean() || fl == L_value(); }
    static constexpr int branch_of(int fl)      // -1: a trunk layer
    {
        for (int b = 0; b < NB; ++b)
            if (fl >= ext(b, 0) && fl < ext(b, 0) + de(b)) return b;
        return -1;
    }
    // VF_ACTIVATION_* of forward layer fl's output: the extractor MLPs' (features_extractor_kwargs.activation_fn), the trunks' (the
    // policy's activation_fn), none on the heads
    static constexpr int act_of(int fl) { return is_head(fl) ? VF_ACTIVATION_NONE : branch_of(fl) >= 0 ? S::EACT : S::ACT; }
    static constexpr int feat_off(int b)
    {
        int n = 0;
        for (int i = 0; i < b; ++i) n += ew(i, de(i) - 1);
        return n;
    }
    static constexpr int n_feat_b() { return feat_off(NB); }          // feature tiles the extractor branches produce
    static constexpr int n_feat() { return feat_off(NB) + PASS; }     // input tiles of the trunks' first layers (+ the pass-through tile)
    static constexpr int t_pass() { return t_feat() + n_feat_b(); }
    // output width of forward layer fl in tiles (heads: 1)
    static constexpr int width(int fl)
    {
        const int b = branch_of(fl);
        if (b >= 0) return ew(b, fl - ext(b, 0));
        if (is_head(fl)) return 1;
        return fl < L_mean() ? S::PW[fl - base()] : S::VW[fl - L_vf(0)];
    }
    // tiles: [extractor hidden outputs, branch by branch][feat = the branches' last outputs][pass][pi hidden ..][mean][vf hidden ..][value]
    static constexpr int t_feat()
    {
        int n = 0;
        for (int b = 0; b < NB; ++b)
            for (int l = 0; l + 1 < de(b); ++l) n += ew(b, l);
        return n;
    }
    static constexpr int tile_of_layer(int fl)
    {
        const int b = branch_of(fl);
        if (b >= 0) {
            const int l = fl - ext(b, 0);
            if (l == de(b) - 1) return t_feat() + feat_off(b);
            int n = 0;
            for (int i = 0; i < b; ++i)
                for (int k = 0; k + 1 < de(i); ++k) n += ew(i, k);
            for (int k = 0; k < l; ++k) n += ew(b, k);
            return n;
        }
        int n = t_feat() + n_feat();
        for (int l = base(); l < fl; ++l) n += width(l);
        return n;
    }
    static constexpr int n_tiles() { return tile_of_layer(L_value()) + 1; }
    // input of forward layer fl: the forward layer that produces it, -1: an observation, -2: the feature concat
    static constexpr int producer(int fl)
    {
        const int b = branch_of(fl);
        if (b >= 0) return fl == ext(b, 0) ? -1 : fl - 1;
        return (fl == base() || fl == L_vf(0)) ? -2 : fl - 1;
    }
    static constexpr int in_tile(int fl) { return producer(fl) == -1 ? 0 : producer(fl) == -2 ? t_feat() : tile_of_layer(fl - 1); }
    static constexpr int in_tiles(int fl) { return producer(fl) == -1 ? S::KIN[branch_of(fl)] / 8 : producer(fl) == -2 ? n_feat() : width(fl - 1); }
    // execution order: the branches level by level, then the two trunks layer by layer -- consecutive layers belong to independent
    // chains, so that one's epilogue (VALU) sits in the shadow of the other's MFMAs (ChainNet::layer)
    static constexpr int exec_desc(int i, bool vf)
    {
        int k = 0;
        for (int l = 0; l < kGenMaxDepth; ++l)
            for (int b = 0; b < NB; ++b)
                if (l < de(b)) {
                    if (k == i) return ext(b, l);
                    ++k;
                }
        for (int j = 0; j <= (DP > DV ? DP : DV); ++j) {
            if (j <= DP) {
                if (k == i) return L_pi(j);
                ++k;
            }
            if (vf && j <= DV) {
                if (k == i) return L_vf(j);
                ++k;
            }
        }
        return -1;
    }
};

template <class N, bool PI, bool VF, bool IG>
struct BwdProgG;

// the tables are built ONCE per class (static constexpr members) and indexed afterwards: the device code asks for layer(i) / op(i) /
// first_item(i) thousands of times per kernel instance, and re-deriving them in the compiler's constexpr interpreter took minutes
constexpr int kGenMaxLayers = 4 * kGenMaxDepth + 2;
struct GenLayers {
    ChainLayer l[kGenMaxLayers];
    int first[kGenMaxLayers + 1];      // first item of layer i; [n] = number of items
    int tile_of[kGenMaxLayers];        // first output tile of forward layer fl
    int n;
};

// PACK: once a layer's input tiles have had their last reader, all the fused kernels' reverse chain wants of them is the ReLU mask --
// they are reduced to one bit per value (ChainLayer::pk0, bit tile = tile number) and their 16 registers each are free again.  The fused
// PPO step of a shape with more than kGenLiveTiles forward tiles runs this variant: with every tile live it spills (26 tiles: 64
// spilled registers, 38 tiles: 283), as bits the whole forward is 0.5 register per tile.
#ifndef VF_GEN_LIVE_TILES
#define VF_GEN_LIVE_TILES 24
#endif
constexpr int kGenLiveTiles = VF_GEN_LIVE_TILES;

template <class S, bool VF, bool PACK = false>
struct GenLayerTable {
    using Shape = GenShape<S>;
    static constexpr GenLayers make()
    {
        GenLayers r{};
        r.n = VF ? Shape::n_layers() : Shape::L_mean() + 1;
        r.first[0] = 0;
        for (int i = 0; i < r.n; ++i) {
            const int fl = Shape::exec_desc(i, VF);
            const int b = Shape::branch_of(fl);
            r.l[i] = ChainLayer{Shape::desc(fl), Shape::producer(fl) == -1 ? b : -1, Shape::in_tile(fl), Shape::in_tiles(fl), Shape::tile_of_layer(fl),
                                Shape::width(fl), Shape::act_of(fl)};
            r.first[i + 1] = r.first[i] + (r.l[i].obs >= 0 ? r.l[i].nin : r.l[i].nin * 4) * r.l[i].nout;
        }
        if (PACK)
            for (int i = 0; i < r.n; ++i) {        // the LAST layer in execution order that reads a tile range packs it
                if (r.l[i].obs >= 0) continue;
                bool last = true;
                for (int k = i + 1; k < r.n; ++k)
                    if (r.l[k].obs < 0 && r.l[k].in0 == r.l[i].in0) last = false;
                if (last) r.l[i].pk0 = r.l[i].in0;
            }
        for (int fl = 0; fl < Shape::n_layers(); ++fl) r.tile_of[fl] = Shape::tile_of_layer(fl);
        return r;
    }
};

template <class S, bool PACK = false>
struct ChainNetG {
    using Shape = GenShape<S>;
    using Spec = S;
    template <bool PI, bool VF2, bool IG>
    using Bwd = BwdProgG<ChainNetG, PI, VF2, IG>;
    static constexpr int NB = S::NB, HV = S::HV, HM = S::HM, PASS = S::PASS;   // heads (4, 1): actor-critic; (4, 4): the SAC-style Actor (mu / log_std); (1, 1) + PASS: its twin critic
    static_assert((HM == 4 && (HV == 1 || HV == 4) && PASS == 0) || (HM == 1 && HV == 1 && PASS == 1), "heads");
    static constexpr bool VF = S::VF;
    static_assert(NB >= 1 && NB <= 2 && S::DP >= 1 && S::DV >= 1 && S::DP <= kGenMaxDepth && S::DV <= kGenMaxDepth, "shape");
    static constexpr int kin(int b) { return S::KIN[b]; }
    static constexpr int base = Shape::base() + PASS;            // layer-table index of the first trunk layer
    static constexpr int n_layers = Shape::n_layers() + PASS;    // layers of the vf_mlp_desc this class matches
    static constexpr int L_ident = Shape::base();                // (PASS) layer-table index of the frozen identity layer
    static constexpr int L_mean = Shape::desc(Shape::L_mean()), L_value = Shape::desc(Shape::L_value());      // layer-table indices of the heads
    static constexpr int n_exec = VF ? Shape::n_layers() : Shape::L_mean() + 1;     // layers the kernel runs
    static constexpr int t_feat = Shape::t_feat(), n_feat = Shape::n_feat();
    static constexpr int t_pass = Shape::t_pass();
    static constexpr int t_mean = Shape::tile_of_layer(Shape::L_mean()), t_val = Shape::tile_of_layer(Shape::L_value());
    static constexpr int n_tiles = Shape::n_tiles();
    static constexpr GenLayers tab = GenLayerTable<S, S::VF, PACK>::make();
    static constexpr bool pack_or = PACK;
    static constexpr int n_mb = PACK ? (Shape::n_tiles() + 1) / 2 : 1;
    static constexpr ChainLayer layer(int i) { return tab.l[i]; }
    static constexpr int groups(int i) { return tab.l[i].obs >= 0 ? tab.l[i].nin : tab.l[i].nin * 4; }
    static constexpr int items(int i) { return tab.first[i + 1] - tab.first[i]; }
    static constexpr int n_items() { return tab.first[n_exec]; }
    static constexpr bool is_head(int i) { return tab.l[i].relu == 0; }
    static constexpr int tile_of_layer(int fl) { return tab.tile_of[fl]; }
    static constexpr int layer_of(int item)
    {
        int i = 0;
        while (item >= tab.first[i + 1]) ++i;
        return i;
    }
    static constexpr int first_item(int li) { return tab.first[li]; }
    static constexpr int act_of(int fl) { return Shape::act_of(fl); }
    static constexpr bool all_relu = S::ACT == VF_ACTIVATION_RELU && S::EACT == VF_ACTIVATION_RELU;
    static_assert(!PACK || all_relu, "activations as bits: ReLU networks only");
    static constexpr int mask_bits(int fl) { return PACK && !Shape::is_head(fl) ? tab.tile_of[fl] : -1; }
};

struct GenOps {
    BwdOp op[kGenMaxOps];
    int first[kGenMaxOps + 1] = {};    // first item of op i; [n] = number of items
    int n = 0;
    int n_ym = 1;
};

template <class N, bool PI, bool VF, bool IG>
struct BwdGenTable {
    using Sh = typename N::Shape;
    static constexpr int g_in(int b) { return N::n_tiles + b; }
    static constexpr GenOps make()
    {
        GenOps r{};
        // the trunks from their heads down, alternating (the two chains are independent until the feature gradient)
        const int n_feat_ops = (PI ? 1 : 0) + (VF ? 1 : 0);
        int seen = 0;
        for (int k = 0; k <= (Sh::DP > Sh::DV ? Sh::DP : Sh::DV); ++k)
            for (int t = 0; t < 2; ++t) {
                if (t == 0 ? !(PI && k <= Sh::DP) : !(VF && k <= Sh::DV)) continue;
                const int fl = t == 0 ? Sh::L_pi(Sh::DP - k) : Sh::L_vf(Sh::DV - k);
                BwdOp o{};
                o.fl = fl;
                o.obs = -1;
                if (Sh::is_head(fl)) {
                    o.in_kind = t == 0 ? 1 : 2;
                    o.in0 = 0;
                    o.G = 1;
                } else {
                    o.in_kind = 0;
                    o.in0 = Sh::tile_of_layer(fl);
                    o.G = Sh::width(fl) * 4;
                }
                if (Sh::producer(fl) == -2) {           // a trunk's first layer: its data gradient is (part of) the feature gradient
                    o.out0 = Sh::t_feat();
                    o.nout = Sh::n_feat_b();              // (the pass-through columns' gradient is not formed: a critic update does not need it)
                    o.accum = seen > 0 ? 1 : 0;
                    ++seen;
                    o.nfin = 0;
                    if (seen == n_feat_ops) {
                        o.nfin = Sh::NB;
                        for (int b = 0; b < Sh::NB; ++b)
                            o.fin[b] = BwdFin{Sh::ext(b, Sh::de(b) - 1), Sh::t_feat() + Sh::feat_off(b), Sh::ew(b, Sh::de(b) - 1), Sh::feat_off(b)};
                    }
                } else {
                    o.out0 = Sh::tile_of_layer(fl - 1);
                    o.nout = Sh::width(fl - 1);
                    o.accum = 0;
                    o.nfin = 1;
                    // the two head ops load their masks in the prologue, both before either is finalised: separate slots
                    const int ym0 = (Sh::is_head(fl) && t == 1 && PI) ? Sh::width(Sh::L_mean() - 1) : 0;
                    o.fin[0] = BwdFin{fl - 1, o.out0, o.nout, ym0};
                }
                r.op[r.n++] = o;
            }
        // extractor layers above the first, level by level
        for (int l = kGenMaxDepth - 1; l >= 1; --l)
            for (int b = 0; b < Sh::NB; ++b) {
                if (l >= Sh::de(b)) continue;
                const int fl = Sh::ext(b, l);
                BwdOp o{};
                o.fl = fl;
                o.in_kind = 0;
                o.in0 = Sh::tile_of_layer(fl);
                o.G = Sh::width(fl) * 4;
                o.out0 = Sh::tile_of_layer(fl - 1);
                o.nout = Sh::width(fl - 1);
                o.accum = 0;
                o.obs = -1;
                o.nfin = 1;
                o.fin[0] = BwdFin{fl - 1, o.out0, o.nout, 0};
                r.op[r.n++] = o;
            }
        if (IG)
            for (int b = 0; b < Sh::NB; ++b) {
                const int fl = Sh::ext(b, 0);
                BwdOp o{};
                o.fl = fl;
                o.in_kind = 0;
                o.in0 = Sh::tile_of_layer(fl);
                o.G = Sh::width(fl) * 4;
                o.out0 = g_in(b);
                o.nout = 1;
                o.accum = 0;
                o.obs = b;
                o.nfin = 0;
                r.op[r.n++] = o;
            }
        for (int i = 0; i < r.n; ++i)
            for (int f = 0; f < r.op[i].nfin; ++f)
                if (r.op[i].fin[f].ym0 + r.op[i].fin[f].nt > r.n_ym) r.n_ym = r.op[i].fin[f].ym0 + r.op[i].fin[f].nt;
        for (int i = 0; i < r.n; ++i) r.first[i + 1] = r.first[i] + r.op[i].G * r.op[i].nout;
        return r;
    }
};

template <class N, bool PI, bool VF, bool IG>
struct BwdProgG {
    using Net = N;
    using Tab = BwdGenTable<N, PI, VF, IG>;
    using Sh = typename N::Shape;
    static_assert(PI || VF, "a reverse chain needs a head gradient");
    static constexpr int NB = N::NB;
    static constexpr bool sac_head = PI && VF && N::HV == 4 && N::HM == 4;     // td_policies.Actor: both heads' gradients can come from d_action (BwdProg)
    static constexpr int L_mean = Sh::L_mean(), L_val = Sh::L_value();       // (MlpPolicy order without the identity layer)
    static constexpr int n_tiles = N::n_tiles + NB;
    static constexpr GenOps tab = Tab::make();
    static constexpr int n_ops = tab.n;
    static constexpr int n_ym = tab.n_ym;
    static constexpr bool included(int l) { return l < Sh::base() || (l <= Sh::L_mean() ? PI : VF); }
    static constexpr int n_entries()
    {
        int e = 0;
        for (int l = 0; l < Sh::n_layers(); ++l) e += included(l) ? 1 : 0;
        return e;
    }
    static constexpr int entry(int fl)
    {
        int e = 0;
        for (int l = fl + 1; l < Sh::n_layers(); ++l) e += included(l) ? 1 : 0;
        return e;
    }
    static constexpr BwdOp op(int i) { return tab.op[i]; }
    static constexpr int items(int i) { return tab.first[i + 1] - tab.first[i]; }
    static constexpr int n_items() { return tab.first[n_ops]; }
    static constexpr int op_of(int item)
    {
        int i = 0;
        while (item >= tab.first[i + 1]) ++i;
        return i;
    }
    static constexpr int first_item(int oi) { return tab.first[oi]; }
};

// ---- does a layer table describe class N?  (chain_matches / bwd_chain_matches of vf_mlp_chain_kernels.hpp, table-driven) ----
template <class N>
bool chain_matches_gen(const vf_mlp_desc& d)
{
    using Sh = typename N::Shape;
    if (d.n_layers != N::n_layers || d.n_inputs != N::NB + N::PASS) return false;
    for (int b = 0; b < N::NB; ++b)
        if (d.in_dim[b] < 1 || ((d.in_dim[b] + 7) & ~7) != N::kin(b)) return false;
    const int fid = d.layer[Sh::ext(0, Sh::de(0) - 1)].dst;
    int pw = 0;
    if constexpr (N::PASS) {      // the frozen identity layer: input NB, <= 4 columns, appended to the features, declared in identity_mask (chain_matches)
        const vf_mlp_layer& I = d.layer[N::L_ident];
        pw = d.in_dim[N::NB];
        if (pw < 1 || pw > 4 || I.K != pw || I.No != pw || I.relu || I.src != N::NB || I.src_col != 0 || I.dst != fid || I.dst_col != 32 * Sh::n_feat_b()) return false;
        if (d.identity_mask != (1 << N::L_ident)) return false;
        if (I.save && ((I.save_ld & 3) || (I.dst_col & 3))) return false;
    } else if (d.identity_mask) {
        return false;
    }
    for (int fl = 0; fl < Sh::n_layers(); ++fl) {
        const vf_mlp_layer& L = d.layer[Sh::desc(fl)];
        const int b = Sh::branch_of(fl), p = Sh::producer(fl);
        const int K = p == -1 ? d.in_dim[b] : p == -2 ? 32 * Sh::n_feat_b() + pw : 32 * Sh::in_tiles(fl);
        const int No = fl == Sh::L_mean() ? N::HM : fl == Sh::L_value() ? N::HV : 32 * Sh::width(fl);
        if (L.K != K || L.No != No || L.relu != Sh::act_of(fl) || L.wr_off < 0 || (L.wr_off & 3)) return false;
        if (p == -1) {
            if (L.src != b || L.src_col != 0) return false;
        } else if (p == -2) {
            if (L.src != fid || L.src_col != 0) return false;
        } else if (L.src != d.layer[Sh::desc(p)].dst || L.src_col != d.layer[Sh::desc(p)].dst_col) {
            return false;
        }
        if (b >= 0 && fl == Sh::ext(b, Sh::de(b) - 1) && (L.dst != fid || L.dst_col != 32 * Sh::feat_off(b))) return false;
        if (b >= 0 && fl != Sh::ext(b, Sh::de(b) - 1) && L.dst_col != 0) return false;
        if (b < 0 && !Sh::is_head(fl) && L.dst_col != 0) return false;
        if (fl == Sh::L_mean() && L.dst != VF_MLP_OUT0) return false;
        if (fl == Sh::L_value() && L.dst != VF_MLP_OUT1) return false;
        if (L.save && ((L.save_ld & 3) || (L.dst_col & 3) || (reinterpret_cast<uintptr_t>(L.save) & 15))) return false;
    }
    return true;
}

template <class P>
bool bwd_chain_matches_gen(const vf_mlp_bwd_desc& d, bool ig)
{
    using N = typename P::Net;
    using Sh = typename N::Shape;
    if (d.n_layers != P::n_entries()) return false;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    for (int fl = 0; fl < Sh::n_layers(); ++fl) {
        if (!P::included(fl)) continue;
        const vf_mlp_bwd_layer& E = d.layer[P::entry(fl)];
        const int b = Sh::branch_of(fl), p = Sh::producer(fl);
        const bool relu = !Sh::is_head(fl);
        if (p == -1) {
            if (E.K < 1 || ((E.K + 7) & ~7) != N::kin(b)) return false;
            if ((E.need_dx != 0) != ig) return false;
        } else if (p == -2 && N::PASS) {      // features (+) pass-through columns: K = feat + pw, pw = 1 .. 4 (bwd_chain_matches)
            if (E.K <= 32 * Sh::n_feat_b() || E.K > 32 * Sh::n_feat_b() + 4 || E.need_dx == 0) return false;
        } else {
            if (E.K != 32 * Sh::in_tiles(fl) || E.need_dx == 0) return false;
        }
        const int No = fl == Sh::L_mean() ? N::HM : fl == Sh::L_value() ? N::HV : 32 * Sh::width(fl);
        if (E.No != No || (E.Y != nullptr) != relu || E.wq_off < 0 || (E.wq_off & 3)) return false;
        if (relu && (E.act ? E.act : VF_ACTIVATION_RELU) != Sh::act_of(fl)) return false;
        if (relu && (!al16(E.Y) || (E.ld_y & 3) || !al16(E.dY) || (E.ld_dy & 3))) return false;
        // wiring: the gradient this layer's weights produce is its producer's dY buffer (the feature gradient: the branch's columns)
        if (p >= 0 && E.dX != d.layer[P::entry(p)].dY) return false;
        if (p == -2) {
            for (int bb = 0; bb < N::NB; ++bb) {
                const vf_mlp_bwd_layer& X = d.layer[P::entry(Sh::ext(bb, Sh::de(bb) - 1))];
                if (X.dY != E.dX + 32 * Sh::feat_off(bb) || X.ld_dy != E.ld_dx) return false;
            }
        }
    }
    return true;
}

// 16 rows per wave for the reverse chain of a generated class?  bwd16_ok's rule (vf_mlp_chain_kernels.hpp)
template <class P>
bool bwd16_ok_gen(const vf_mlp_bwd_desc& d, int M)
{
    using Sh = typename P::Net::Shape;
    static const int forced = [] { const char* e = getenv("VISFLY_AMD_MLP_CHAIN16"); return e ? atoi(e) : -1; }();
    if (forced == 0 || (forced < 0 && M > 16384)) return false;
    for (int l = 0; l < d.n_layers; ++l)
        if (d.layer[l].wb_off < 0) return false;
    for (int b = 0; b < Sh::NB; ++b)
        if (d.layer[P::entry(Sh::ext(b, 0))].K > 16) return false;
    return true;
}

}  // namespace vf
