// vf_mlp_chain.hpp -- shared definitions of the register-chained MLP kernels (network classes, launch arguments) and the
// 16-rows-per-wave forward chain as device code, so that other translation units can run the policy forward inside their own
// kernels (vf_bptt_rollout.hip: policy forward + env step of a whole BPTT horizon in one persistent launch).  The kernels
// themselves, the 32-row chains and the reverse chains live in vf_mlp_chain.hip.
#pragma once
#include "vf_common.hpp"
#include "vf_ppo_device.hpp"

#include <type_traits>

#ifndef VF_CHAIN16_WT
#define VF_CHAIN16_WT 2      // A fragments of the 16-row forward: 2 = float4 of the 32-row chain's image (hidden layers), 1 = four dwords of the
                             // transposed image, 0 = float4 of the row-major parameter buffer (A/B)
#endif
#ifndef VF_CHAIN16_DEPTH
#define VF_CHAIN16_DEPTH 24
#endif

#ifndef VF_CHAIN_HOOK
#define VF_CHAIN_HOOK(kind, idx) do { } while (0)     // trace builds: a cycle stamp at the end of layer / op idx (kind 0 / 1: forward before / after the epilogue, 2 / 3: reverse)
#endif

namespace vf {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct __attribute__((packed, aligned(4))) f32x4u {   // 4 consecutive floats at dword alignment (bias vectors)
    float x, y, z, w;
};

// pointers that went through an opaque register copy (chain_store_setup) lose the compiler's "this is global memory" inference and
// would be dereferenced with flat_* instructions (64-bit VGPR addresses, lgkmcnt as well as vmcnt): say it in the type
typedef __attribute__((address_space(1))) char* vf_gptr;
typedef float vf_st4 __attribute__((ext_vector_type(4)));              // (float4 is a class: no assignment operator for address space 1)
typedef __attribute__((address_space(1))) vf_st4 vf_gfloat4;

struct ChainIo {
    const float* in[3];          // observation inputs of the extractor branches, then the pass-through input (if the class has one)
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
    // ---- slices (vf_mlp_chain_split.hpp: two waves share a row tile, each walks half of the network); defaults = a whole layer ----
    int a0 = 0;            // the layer's output tiles [a0, a0 + nout) are computed here (weight image / bias / saved-copy columns)
    int s0 = 0, sn = -1;   // of those, tiles [s0, s0 + sn) are the ones THIS wave stores for the weight gradients (-1: all)
    int xsn = 0, xr0 = 0;  // > 0: after the epilogue the output tiles go to the partner wave through LDS and its xsn tiles arrive at xr0
    int pk0 = -1;          // >= 0: at the end of this layer its INPUT tiles are reduced to one bit per value (> 0?) at bit tiles pk0 .. of
                           // ChainState::mb -- all the fused kernels' reverse chain wants of them -- and the 16 registers per tile are free
};

// NB branches (KA / KB = first-layer widths padded to 8), hidden widths in 32-feature tiles
// VF_ = false: the value trunk is not executed (first-order policy optimisation only needs the action mean)
// HV_ / HM_: widths of the second / first trunk's head -- (HM, HV) = (4, 1): actor-critic of the PPO policies (policies.py:18-49);
//   (4, 4): the reference's SAC-style Actor (utils/policies/td_policies.py:146-252: latent_pi -> mu, log_latent_pi -> log_std);
//   (1, 1): its twin ContinuousCritic (:82-143: qf0 / qf1 -> Q)
// PASS_ = 1: one more input (<= 8 wide; the critic's action) is appended to the features unchanged -- th.cat([features, actions])
//   (:137).  In the layer table it is a frozen identity layer between the extractor layers and the trunks (vf_mlp_desc.identity_mask);
//   here it is one more 32-feature input tile of the trunks' first layers whose first registers hold the input's columns
template <class N, bool PI, bool VF, bool IG>
struct BwdProg;      // vf_mlp_chain_bwd.hpp

template <int NB_, int KA_, int KB_, int E1_, int E2_, int P1_, int P2_, int V1_, int V2_, bool VF_ = true, int HV_ = 1, int HM_ = 4,
          int PASS_ = 0>
struct ChainNet {
    template <bool PI, bool VF2, bool IG>
    using Bwd = BwdProg<ChainNet, PI, VF2, IG>;      // the class's reverse-chain program (a generated class names its own: vf_mlp_chain_gen.hpp)
    static constexpr int NB = NB_, E1 = E1_, E2 = E2_, P1 = P1_, P2 = P2_, V1 = V1_, V2 = V2_, HV = HV_, HM = HM_, PASS = PASS_;
    static constexpr bool VF = VF_;
    static_assert((HV_ == 1 || HV_ == 4) && (HM_ == 1 || HM_ == 4), "heads: 1 or 4 wide");
    static_assert(PASS_ == 0 || PASS_ == 1, "at most one pass-through input");
    static constexpr int kin(int b) { return b == 0 ? KA_ : KB_; }
    static constexpr int base = 2 * NB + PASS;               // desc index of the first trunk layer
    static constexpr int L_ident = 2 * NB;                   // (PASS) desc index of the frozen identity layer
    static constexpr int n_layers = base + 6;                // layers of the vf_mlp_desc this class matches
    static constexpr int n_exec = 2 * NB + (VF ? 6 : 3);     // layers the kernel runs (the identity layer is not one of them)
    static constexpr int L_mean = base + 2, L_value = base + 5;   // desc indices of the heads
    // tiles: [branch L1 outputs][feat][pass][pi1][pi2][mean][vf1][vf2][value]
    static constexpr int t_e1(int b) { return b * E1; }
    static constexpr int t_feat = NB * E1;
    static constexpr int t_pass = t_feat + NB * E2;
    static constexpr int t_p1 = t_pass + PASS, t_p2 = t_p1 + P1, t_mean = t_p2 + P2;
    static constexpr int t_v1 = t_mean + 1, t_v2 = t_v1 + V1, t_val = t_v2 + V2;
    static constexpr int n_tiles = t_val + 1;
    static constexpr int n_feat = NB * E2 + PASS;            // input tiles of the trunks' first layers
    // execution order alternates between independent chains so that one chain's epilogue (VALU) can sit in the
    // shadow of the other's MFMAs
    static constexpr ChainLayer layer(int i)
    {
        if (i < NB) return ChainLayer{2 * i, i, 0, kin(i) / 8, t_e1(i), E1, 1};
        if (i < 2 * NB) return ChainLayer{2 * (i - NB) + 1, -1, t_e1(i - NB), E1, t_feat + (i - NB) * E2, E2, 1};
        if (!VF) {
            switch (i - 2 * NB) {
            case 0: return ChainLayer{base + 0, -1, t_feat, n_feat, t_p1, P1, 1};
            case 1: return ChainLayer{base + 1, -1, t_p1, P1, t_p2, P2, 1};
            default: return ChainLayer{base + 2, -1, t_p2, P2, t_mean, 1, 0};
            }
        }
        switch (i - 2 * NB) {
        case 0: return ChainLayer{base + 0, -1, t_feat, n_feat, t_p1, P1, 1};
        case 1: return ChainLayer{base + 3, -1, t_feat, n_feat, t_v1, V1, 1};
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
        for (int i = 0; i < n_exec; ++i) n += items(i);
        return n;
    }
    static constexpr bool is_head(int i) { return layer(i).desc == L_mean || layer(i).desc == L_value; }
    // first output tile of forward layer fl in MlpPolicy order WITHOUT the identity layer (the numbering of BwdProg)
    static constexpr int tile_of_layer(int fl)
    {
        if (fl < 2 * NB) return (fl & 1) ? t_feat + (fl >> 1) * E2 : t_e1(fl >> 1);
        switch (fl - 2 * NB) {
        case 0: return t_p1;
        case 1: return t_p2;
        case 2: return t_mean;
        case 3: return t_v1;
        case 4: return t_v2;
        default: return t_val;
        }
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
    static constexpr int act_of(int /*fl*/) { return VF_ACTIVATION_RELU; }     // activation of forward layer fl's output (the built-in classes: ReLU networks)
    static constexpr int mask_bits(int /*fl*/) { return -1; }     // first bit tile of forward layer fl's ReLU mask (ChainLayer::pk0), -1: the tiles stay
    static constexpr int n_mb = 8;              // words of ChainState::mb
    static constexpr bool pack_or = false;      // true: bit tiles are packed in any order into words the kernel zeroed (vf_mlp_chain_gen.hpp)
};

constexpr int kChainDepth = 8;   // weight blocks in flight: 8 x 4 MFMAs x 64 cycles = 2 k cycles of cover (4 deep measured the same in the split kernels)

template <class N>
struct ChainState {
    using Net = N;
    f32x16 t[N::n_tiles];
    unsigned mb[N::n_mb];        // ReLU masks kept as bits (ChainLayer::pk0): bit tile j = bits [16 (j & 1), +16) of word j >> 1, bit r = register r
    float x[2][16];              // observation fragments: x[b][s] = X[m][2 s + h] (K padded to <= 32)
    float4 ring[kChainDepth];
    float4 bias[4][4];           // bias of the layer in flight: [out tile][g] -> features 32 a + 8 g + 4 h .. + 3
    // where the saved copy of the PREVIOUS layer's output goes (chain_store_setup): wave-uniform base (null: not kept) + this lane's byte offset
    unsigned long sv_base;       // (0: not kept)
    unsigned sv_off;
};

struct ChainArgs {
    vf_mlp_desc d;
    const float* params;
    const float* packed;
    ChainIo io;
    int M;
    // optional action head (vf_mlp_forward_act): action = tanh(mean + exp(log_std) * eps) instead of the mean
    const float* rp_log_std;
    const float4* rp_eps;
    float4* rp_action;
    float* obs_copy[2];      // optional: the observation rows are also written here (a trainer's contiguous per-slot copy)
    // HV == 4 classes (td_policies.Actor: the second head IS log_std, rp_log_std unused): clamp bounds of the state-dependent
    // log_std in action = tanh(mean + eps exp(clamp(log_std, lo, hi))) (k_shac_head_fwd's arithmetic)
    float rp_ls_lo, rp_ls_hi;
    // a persistent caller's exp(rp_log_std[k]), computed once per launch (rp_std_valid != 0), instead of four loads + expf per pass
    int rp_std_valid;
    float rp_std[4];
};

__device__ __forceinline__ float chain_clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

using NetHover = ChainNet<1, 16, 8, 4, 2, 2, 2, 2, 2>;   // StateExtractor [128, 64], pi / vf [64, 64]
using NetNav = ChainNet<2, 16, 8, 4, 2, 2, 2, 2, 2>;     // StateTargetExtractor [128, 64] x 2, pi / vf [64, 64]
using NetHoverPi = ChainNet<1, 16, 8, 4, 2, 2, 2, 2, 2, false>;
using NetNavPi = ChainNet<2, 16, 8, 4, 2, 2, 2, 2, 2, false>;
using NetSacHover = ChainNet<1, 16, 8, 4, 2, 2, 2, 2, 2, true, 4>;   // td_policies.Actor over StateExtractor: mu / log_std heads
using NetSacNav = ChainNet<2, 16, 8, 4, 2, 2, 2, 2, 2, true, 4>;     // ... over StateTargetExtractor
using NetCriticHover = ChainNet<1, 16, 8, 4, 2, 2, 2, 2, 2, true, 1, 1, 1>;   // td_policies.ContinuousCritic: extractor, (+) action, qf0 / qf1


// ---- the 32-rows-per-wave forward (the scheme at the head of vf_mlp_chain.hip) ----
// The weight fragments are BUFFER loads: descriptor over the packed image (4 SGPRs, made once) + this lane's constant 16-byte slot +
// a scalar block offset.  r05 (tools/mfma_occupancy_probe.hip, profiles/r05_chain_split.txt): what a load costs a lone wave is its
// ISSUE, during which the wave's dependent MFMAs cannot go -- 49 cycles per global_load with a 64-bit VGPR address (what the pointer
// form below compiled to, plus two VALU adds), 71 with scalar base + VGPR offset, 30 as a buffer load, per item of 4 MFMAs = 256 cycles
#ifndef VF_CHAIN_BUFFER_LOADS
#define VF_CHAIN_BUFFER_LOADS 1
#endif

typedef unsigned vf_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t chain_weight_rsrc(const float* packed)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(packed), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ float4 chain_buffer_float4(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned block_off)
{
    const vf_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)block_off, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}

// ... and the trickled copies are BUFFER stores: a global_store with scalar base + lane offset costs a lone wave 26 cycles of issue, with a
// 64-bit VGPR address 32, a buffer_store 2 (same probe).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t chain_store_rsrc(unsigned long base)
{
    // (readfirstlane: the pinned base went through an inline asm, whose result the compiler takes for divergent -- it would wrap every
    // store in a waterfall loop over "different" descriptors.  For the same reason the descriptor has a constant size and a buffer that
    // is not kept is skipped by a scalar branch: `base ? 2^32 - 1 : 0` becomes a v_cndmask however it is written)
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base), hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long)hi << 32) | lo), 0, 0xFFFFFFFFu, 0x00020000);
}
// The column offset (a compile-time constant < 4096) goes into the instruction's IMMEDIATE offset field (voffset + constant, soffset = 0),
// NOT into soffset.  r05, found on a generated class: with the constant in soffset, offsets above 64 (no inline constant) become
// `s_movk_i32 sN, 0xc0; buffer_store_dwordx4 v[6:9], v122, s[28:31], sN offen; v_add_f32 v6, ..` -- the compiler's hazard rule for
// 128-bit store data ("a VALU write of the data registers needs 2 wait states after the store") exempts stores whose soffset is an SGPR,
// and the next layer's epilogue overwrote v6 in the very next slot.  On gfx950 the exemption does not hold when the store's issue stalls
// (first touch of freshly allocated pages): element 0 of the float4 of the last-read lanes (12-15 of every 16) went out as the NEW value
// of v6, a pre-ReLU sum -- 8 rows x 2 floats of one saved activation wrong, once in ~3 cold launches.  With the immediate form the
// compiler sees a store without soffset register and inserts the `s_nop 1` itself.
#ifndef VF_CHAIN_STORE_AUX
#define VF_CHAIN_STORE_AUX 0      // cache policy of the saved-activation / masked-gradient stores (bit 0 sc0, bit 1 nt, bit 4 sc1): A/B knob, profiles/r06_fused_tail.txt
#endif
__device__ __forceinline__ void chain_buffer_store(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned col_off, float a, float b, float c, float d)
{
    __builtin_amdgcn_raw_buffer_store_b128(vf_u4{__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)}, r, (int)(lane_off + col_off), 0, VF_CHAIN_STORE_AUX);
}

template <class N, int I>
__device__ __forceinline__ float4 chain_load(const ChainArgs& g, int lane)
{
    constexpr int li = N::layer_of(I), local = I - N::first_item(li);
    constexpr ChainLayer L = N::layer(li);
    constexpr int G = N::groups(li), gq = local / L.nout, a = local % L.nout;
#if VF_CHAIN_BUFFER_LOADS
    return chain_buffer_float4(chain_weight_rsrc(g.packed), (unsigned)lane * 16u, (unsigned)g.d.layer[L.desc].wr_off * 4u + ((L.a0 + a) * G + gq) * 1024u);
#else
    // wave-uniform base (scalar registers) + 32-bit lane offset: no 64-bit VGPR address arithmetic per load
    const char* base = reinterpret_cast<const char*>(g.packed + g.d.layer[L.desc].wr_off) + ((L.a0 + a) * G + gq) * 1024;
    return *reinterpret_cast<const float4*>(base + (unsigned)lane * 16u);
#endif
}

// widths are compile-time (hidden layers: whole tiles; heads: 4 / 1 features in lane half 0, q = 0), so the bias loads
// and the epilogue carry no guards
template <class N, int LI>
__device__ __forceinline__ void chain_bias_load(const ChainArgs& g, ChainState<N>& st, int h)
{
    constexpr ChainLayer L = N::layer(LI);
    const float* b = g.params + g.d.layer[L.desc].b_off;    // b_off is only dword aligned
    if constexpr ((L.desc == N::L_value && N::HV == 1) || (L.desc == N::L_mean && N::HM == 1)) {
        st.bias[0][0] = make_float4(b[0], 0.0f, 0.0f, 0.0f);
    } else if constexpr (L.desc == N::L_mean || L.desc == N::L_value) {
        const f32x4u v = *reinterpret_cast<const f32x4u*>(b);
        st.bias[0][0] = make_float4(v.x, v.y, v.z, v.w);
    } else {
#pragma unroll
        for (int a = 0; a < L.nout; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4u v = *reinterpret_cast<const f32x4u*>(b + 32 * (L.a0 + a) + 8 * q + 4 * h);
                st.bias[a][q] = make_float4(v.x, v.y, v.z, v.w);
            }
    }
}

template <class N, int LI>
__device__ __forceinline__ void chain_epilogue(const ChainArgs& g, ChainState<N>& st, int row, int h, bool live)
{
    constexpr ChainLayer L = N::layer(LI);
    if constexpr (N::is_head(LI)) {                    // heads: mean (M,4) / value (M,1); only lane half 0 holds them
        f32x16& y = st.t[L.out0];
        const float4 bq = st.bias[0][0];
        y[0] += bq.x; y[1] += bq.y; y[2] += bq.z; y[3] += bq.w;
        if (live && h == 0) {                          // (the fused PPO kernel keeps the heads in registers: no pointers)
            if constexpr (L.desc == N::L_mean && N::HM == 1) {
                if (g.io.mean) g.io.mean[row] = y[0];
            } else if constexpr (L.desc == N::L_mean) {
                if (g.io.mean) *reinterpret_cast<float4*>(g.io.mean + (size_t)row * 4) = make_float4(y[0], y[1], y[2], y[3]);
                if constexpr (N::HV != 4) {
                    if (g.rp_action) {     // k_reparam_fwd's arithmetic on the head still in registers
                        const float4 e = g.rp_eps[row];
                        g.rp_action[row] = make_float4(tanhf(y[0] + expf(g.rp_log_std[0]) * e.x), tanhf(y[1] + expf(g.rp_log_std[1]) * e.y),
                                                       tanhf(y[2] + expf(g.rp_log_std[2]) * e.z), tanhf(y[3] + expf(g.rp_log_std[3]) * e.w));
                    }
                }
            } else if constexpr (N::HV == 4) {
                if (g.io.value) *reinterpret_cast<float4*>(g.io.value + (size_t)row * 4) = make_float4(y[0], y[1], y[2], y[3]);
                if (g.rp_action) {         // k_shac_head_fwd's arithmetic: the mean head (run before this one) is still in its tile
                    const f32x16& mu = st.t[N::t_mean];
                    const float4 e = g.rp_eps[row];
                    const float lo = g.rp_ls_lo, hi = g.rp_ls_hi;
                    g.rp_action[row] = make_float4(tanhf(mu[0] + e.x * expf(chain_clampf(y[0], lo, hi))), tanhf(mu[1] + e.y * expf(chain_clampf(y[1], lo, hi))),
                                                   tanhf(mu[2] + e.z * expf(chain_clampf(y[2], lo, hi))), tanhf(mu[3] + e.w * expf(chain_clampf(y[3], lo, hi))));
                }
            } else {
                if (g.io.value) g.io.value[row] = y[0];
            }
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
            for (int r = 0; r < 16; ++r) y[r] = act_fwd_c<L.relu>(y[r]);
        }
    }
}

// The copies of a layer's output that the backward reads are not stored in the epilogue (a 16 KiB burst per wave, all
// waves in lock-step, behind which the weight loads of the following items would queue: loads and stores retire in
// order on gfx9's vmcnt) but trickled out, a float4 or two per item of the NEXT layer in execution order.
//
// r04: what a store needs of the layer table -- base pointer, row stride -- is read ONCE per layer (chain_store_setup at the layer's
// first item) and pinned in registers.  Read per store, hipcc re-loaded the two fields from the kernel-argument segment in front of
// every store (s_load_dword x 2 + s_waitcnt lgkmcnt(0) + v_mul_lo_u32: the scalar-cache round trip fully exposed, ~100 cycles of a
// lone wave per store; 357 s_load in k_ppo_update_chain), and the `row < M` guard made every store an exec-masked branch.  The lanes
// past the last row are exact replicas of row M - 1 (they load THAT row's inputs), so they store the same values to the same
// addresses and need no guard.
template <class N, int LI>
__device__ __forceinline__ void chain_store_setup(const ChainArgs& g, ChainState<N>& st, int rc, int h)
{
    if constexpr (LI >= 1 && !N::is_head(LI >= 1 ? LI - 1 : 0)) {
        constexpr ChainLayer P = N::layer(LI - 1);
        const vf_mlp_layer& D = g.d.layer[P.desc];
        unsigned long b = D.save ? reinterpret_cast<unsigned long>(D.save + D.dst_col) : 0ul;
        asm volatile("" : "+s"(b));          // opaque: the compiler cannot re-derive it from the kernel arguments at every use
        st.sv_base = b;
        // lane offset in BYTES, 32 bit (the hosts refuse row counts whose buffers pass 4 GiB)
        st.sv_off = ((unsigned)rc * (unsigned)D.save_ld + 4u * h) * 4u;
    }
}

template <class N, int LI, int LOCAL>
__device__ __forceinline__ void chain_deferred_store(const ChainState<N>& st)
{
    if constexpr (LI >= 1 && !N::is_head(LI >= 1 ? LI - 1 : 0)) {
        constexpr ChainLayer P = N::layer(LI - 1);
        constexpr int S = (P.sn < 0 ? P.nout : P.sn) * 4, per = (S + N::items(LI) - 1) / N::items(LI);
        constexpr int s0 = LOCAL * per, s1 = (LOCAL + 1) * per < S ? (LOCAL + 1) * per : S;
        if constexpr (s0 < s1) {
            if (st.sv_base) {
                const __amdgpu_buffer_rsrc_t r = chain_store_rsrc(st.sv_base);
#pragma unroll
                for (int i = s0; i < s1; ++i) {
                    const int a = P.s0 + i / 4, q = i % 4;
                    const f32x16& y = st.t[P.out0 + a];
                    chain_buffer_store(r, st.sv_off, (32 * (P.a0 + a) + 8 * q) * 4, y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
                }
            }
        }
    }
}

// ---- hand-over between the two waves of a split row tile (vf_mlp_chain_split.hpp): same lane -> same lane, one float4 per (tile, q) ----
constexpr int kXchTiles = 2;
__shared__ vf_st4 vf_xch_fwd[2][kXchTiles][4][64];     // [writer role][tile][q][lane]: 16 KiB, allocated only for kernels that exchange
__shared__ vf_st4 vf_xch_bwd[2][kXchTiles][4][64];

// LDS writes of this wave done, then the workgroup barrier.  NOT __syncthreads(): its fence also waits for every global load / store
// in flight (vmcnt(0)), i.e. for the weight ring and the trickled stores
__device__ __forceinline__ void xch_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <class N, int LI>
__device__ __forceinline__ void chain_exchange(ChainState<N>& st, int lane)
{
    constexpr ChainLayer L = N::layer(LI);
    if constexpr (L.xsn > 0) {
        static_assert(L.xsn <= kXchTiles, "exchange buffer");
        constexpr int R = N::role;
#pragma unroll
        for (int a = 0; a < L.xsn; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x16& y = st.t[L.out0 + a];
                vf_xch_fwd[R][a][q][lane] = vf_st4{y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]};
            }
        xch_barrier();
#pragma unroll
        for (int a = 0; a < L.xsn; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const vf_st4 v = vf_xch_fwd[1 - R][a][q][lane];
                f32x16& y = st.t[L.xr0 + a];
                y[4 * q] = v[0]; y[4 * q + 1] = v[1]; y[4 * q + 2] = v[2]; y[4 * q + 3] = v[3];
            }
    }
}

// y > 0 of a ReLU output (>= 0, so: bits != 0) as one bit per accumulator register: 2 VALU per value to pack, 2 to apply (bwd_finalize)
template <class N, int LI>
__device__ __forceinline__ void chain_pack_input(ChainState<N>& st)
{
    constexpr ChainLayer L = N::layer(LI);
    if constexpr (L.pk0 >= 0) {
#pragma unroll
        for (int a = 0; a < L.nin; ++a) {
            const int j = L.pk0 + a;
            unsigned bits = (N::pack_or || (j & 1)) ? st.mb[j >> 1] : 0u;
            const f32x16& y = st.t[L.in0 + a];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float yr = y[r];          // (a copy: __builtin_bit_cast applied to a vector element reads element 0 whatever r is)
                const unsigned nz = __float_as_uint(yr) != 0u ? 1u : 0u;
                bits |= nz << (16 * (j & 1) + r);
            }
            // pinned HERE: left to itself the compiler sinks the 2 x 16 instructions to the first use of the word -- the reverse chain's
            // finalize, a forward trunk and a loss later -- and the tile's 16 registers stay live all the way (measured: 64 VGPRs)
            asm volatile("" : "+v"(bits));
            st.mb[j >> 1] = bits;
        }
    }
}

// SAVE = false: the caller's layer table keeps no activation copies (inference inside a persistent launch): the trickled stores
// and their per-item `save != null` branches are not compiled in (2.5 k cycles of 62 k per forward in k_ppo_rollout)
// rc = the row this lane LOADS (min(row, M - 1): lanes past the last row replicate it), row = the row it would own
template <class N, int I, bool SAVE = true>
__device__ __forceinline__ void chain_items(const ChainArgs& g, ChainState<N>& st, int lane, int row, bool live, int rc)
{
    if constexpr (I < N::n_items()) {
        constexpr int li = N::layer_of(I), local = I - N::first_item(li);
        constexpr ChainLayer L = N::layer(li);
        constexpr int gq = local / L.nout, a = local % L.nout;
        const int h = lane >> 5;
        const float4 w = st.ring[I % kChainDepth];       // the slot being refilled is the one this item consumes: read it first
        if constexpr (I + kChainDepth < N::n_items()) st.ring[I % kChainDepth] = chain_load<N, I + kChainDepth>(g, lane);
        if constexpr (local == 0) chain_bias_load<N, li>(g, st, h);
        f32x16& acc = st.t[L.out0 + a];
        if constexpr (gq == 0) acc = f32x16{0};
        // the refill load and its address arithmetic issue HERE, between the previous item's MFMAs and this item's (different
        // accumulators), not between two MFMAs of this item: an extra issue slot between MFMAs on the SAME accumulator costs
        // ~43 cycles (MI355X_MICROARCH.md, per-instruction constants), and the scheduler liked to put them behind the first one
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float b;
            if constexpr (L.obs >= 0) b = st.x[L.obs][4 * gq + j];
            else b = st.t[L.in0 + gq / 4][4 * (gq % 4) + j];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(j == 0 ? w.x : j == 1 ? w.y : j == 2 ? w.z : w.w, b, acc, 0, 0, 0);
        }
        if constexpr (SAVE && local == 0) chain_store_setup<N, li>(g, st, rc, h);
        if constexpr (SAVE) chain_deferred_store<N, li, local>(st);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (local == N::items(li) - 1) {
            VF_CHAIN_HOOK(0, li);
            chain_epilogue<N, li>(g, st, row, h, live);
            chain_exchange<N, li>(st, lane);
            chain_pack_input<N, li>(st);
            VF_CHAIN_HOOK(1, li);
        }
        chain_items<N, I + 1, SAVE>(g, st, lane, row, live, rc);
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

// the pass-through input as one more activation tile (ChainNet::PASS): columns 0 .. pw-1 of row `rc` of the extra input are features
// 0 .. pw-1 of tile t_pass -- an accumulator lane (m, h) holds features 4 h + (r & 3) + 8 (r >> 2) in register r, i.e. the half
// h = 0 holds them in registers 0 .. 3 (pw <= 4) and 8 .. 11 of h = 0 would be features 8 ..; everything else is zero.  Also the copy
// the weight gradients of the trunks' first layers read: the identity layer's columns of the saved feature rows
template <class N, bool STORE = true>
__device__ __forceinline__ void chain_pass_tile(const ChainArgs& g, ChainState<N>& st, int row, int rc, int h, bool live)
{
    if constexpr (N::PASS) {
        const int pw = g.d.in_dim[N::NB];
        const float* x = g.io.in[N::NB] + (size_t)rc * pw;
        f32x16& t = st.t[N::t_pass];
        t = f32x16{0};
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (h == 0 && k < pw) ? x[k < pw ? k : pw - 1] : 0.0f;
        t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
        const vf_mlp_layer& D = g.d.layer[N::L_ident];
        if (STORE && D.save && live && h == 0) {
            float* o = D.save + (size_t)row * D.save_ld + D.dst_col;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < pw) o[k] = v[k];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same forward with 16 rows per wave (v_mfma_f32_16x16x4_f32) for SMALL row counts.  The 32-row chain puts M / 32 waves on
// 1 024 SIMDs and lasts as long as one wave needs for the whole network whatever M is; at the BPTT shard (16 384 rows = 512
// waves) half of the chip idles.  With 16 rows a wave does half the work and twice as many waves run (probe, same skeleton:
// 47 -> 28 us at 16 384 rows; no gain once the chip is full -- profiles/r02_mfma_chain_probe.txt), so vf_mlp_forward picks
// this kernel for M <= 16 384.
//     A operand = weights   A[i = n][k]    lane = n + 16 kq
//     B operand = X^T       B[k][j = m]    lane = m + 16 kq
//     C / D     = Y^T       D[i = n][j = m] lane (m, gq = lane >> 4) holds n = 4 gq + r, r = 0..3
// An accumulator lane holds features 4 gq .. 4 gq + 3 of its own row: as the B operand of step r of the next layer it supplies
// k = (feature 4 gq + r of that 16-feature tile), so step r's A fragment is W[n][16 T + 4 gq + r] -- the four steps of an
// (output tile, input tile) pair are ONE float4 of the row-major weight matrix itself -- which is how they were read until a
// look at the access pattern: a quarter-wave of that load is 16 lanes n with 16 different rows, 16 B out of each of 16 lines.
// Hidden layers now take the float4 from the 32-row chain's own image, where the same four values of lane (n, kq) sit in lane
// order (chain16_load: quarter-waves of 256 contiguous bytes); observation layers four dwords of the transposed image of the
// block-tile forward (Wt[k][n], vf_mlp_layer.wt_off; quarter-waves of 64 contiguous bytes).  A quarter of the L1 line accesses:
// -3.2 us per step in k_bptt_rollout.
using f32x4 = __attribute__((ext_vector_type(4))) float;

#ifdef VF_CHAIN_TRACE
__device__ long long vf_chain_trace[2][32];
#define VF_TRACE(k) do { if ((blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && threadIdx.x == 0) vf_chain_trace[blockIdx.x ? 1 : 0][k] = __builtin_readcyclecounter(); } while (0)
#else
#define VF_TRACE(k) do { } while (0)
#endif

// weight fragments in flight: an item is 4 MFMAs x 32 cycles here, so the same ~2 k cycles of cover need more slots than the
// 32-row chain's 8 (measured: 8 slots 17.9 us, i.e. load-latency bound)
constexpr int kChain16Depth = VF_CHAIN16_DEPTH;

template <class N>
struct Chain16 {
    static constexpr int nin(int i) { return N::layer(i).obs >= 0 ? 1 : 2 * N::layer(i).nin; }       // 16-feature input tiles
    static constexpr int nout(int i) { return N::is_head(i) ? 1 : 2 * N::layer(i).nout; }
    static constexpr int items(int i) { return nin(i) * nout(i); }
    static constexpr int n_items()
    {
        int n = 0;
        for (int i = 0; i < N::n_exec; ++i) n += items(i);
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

template <class N>
struct ChainState16 {
    f32x4 t[2 * N::n_tiles];
    float x[2][4];               // observation fragment: x[b][j] = X[m][4 gq + j] (K <= 16)
    float4 ring[kChain16Depth];
    float4 bias[8];              // bias of the layer in flight: [out tile] -> features 16 a + 4 gq .. + 3
    vf_gptr sv_base;         // saved copy of the previous layer's output (chain16_store_setup): uniform base (null: not kept) + lane byte offset
    unsigned sv_off;
    float4 act;              // the action the head epilogue wrote to rp_action (lane group 0; k_bptt_rollout hands it on through LDS)
};

template <class N, int I>
__device__ __forceinline__ float4 chain16_load(const ChainArgs& g, int lane)
{
    using C = Chain16<N>;
    constexpr int li = C::layer_of(I), local = I - C::first_item(li);
    constexpr ChainLayer L = N::layer(li);
    constexpr int T = local / C::nout(li), a = local % C::nout(li);
    const vf_mlp_layer& D = g.d.layer[L.desc];
#if VF_CHAIN16_WT == 2
    if constexpr (L.obs < 0) {
        // ONE float4 of the 32-row chain's own image (vf_mlp_layer.wr_off, block (a, g) = 1 KiB, lane l = n + 32 h holds
        // W[32 a + n][32 (g >> 2) + 8 (g & 3) + 4 h .. + 3]): the fragment of lane (i, kq) for output 16-tile a16, input 16-tile T is the
        // float4 of lane 16 (a16 & 1) + i + 32 (kq & 1) in block (a16 / 2, 4 (T >> 1) + 2 (T & 1) + (kq >> 1)); a quarter-wave reads
        // 256 contiguous bytes
        constexpr int G = N::groups(li), blk = (a >> 1) * G + 4 * (T >> 1) + 2 * (T & 1);
        const unsigned kq = lane >> 4;
        // (plain loads: as buffer loads these measured no gain in the persistent launches -- profiles/r05_chain_split.txt section 5)
        const char* base = reinterpret_cast<const char*>(g.packed + D.wr_off) + (blk * 1024 + 256 * (a & 1));       // wave-uniform
        return *reinterpret_cast<const float4*>(base + ((kq >> 1) * 1024u + (kq & 1u) * 512u + (unsigned)(lane & 15) * 16u));
    }
#endif
#if VF_CHAIN16_WT
    {   // four dwords of the zero-padded TRANSPOSED image (vf_mlp_layer.wt_off: Wt[k][n], rows of N32 floats): a quarter-wave (16
        // lanes n, one k) reads 64 contiguous bytes.  The float4 W[n][16 T + 4 gq ..] of the row-major parameters is one
        // instruction instead of four, but its quarter-waves touch 16 rows each: 64 line accesses per item instead of 16
        constexpr int N32 = N::is_head(li) ? 32 : 32 * L.nout;
        const char* base = reinterpret_cast<const char*>(g.packed + D.wt_off + (16 * T * N32 + 16 * a));     // wave-uniform
        const float* p = reinterpret_cast<const float*>(base + ((unsigned)(lane >> 4) * (4u * N32) + (unsigned)(lane & 15)) * 4u);
        return make_float4(p[0], p[N32], p[2 * N32], p[3 * N32]);
    }
#endif
    const int n = 16 * a + (lane & 15), gq = lane >> 4;
    const float* w = g.params + D.w_off;
    if constexpr (L.obs >= 0) {          // K = in_dim <= 16, rows not 16-byte aligned: guarded scalar loads
        const float* r = w + (size_t)n * D.K;
        const int k = 4 * gq;
        return make_float4(k < D.K ? r[k] : 0.0f, k + 1 < D.K ? r[k + 1] : 0.0f, k + 2 < D.K ? r[k + 2] : 0.0f, k + 3 < D.K ? r[k + 3] : 0.0f);
    } else {
        const int nc = n < D.No ? n : D.No - 1;      // heads: rows past No repeat the last one (their outputs are never read)
        return *reinterpret_cast<const float4*>(w + (size_t)nc * D.K + 16 * T + 4 * gq);
    }
}

template <class N, int LI>
__device__ __forceinline__ void chain16_bias_load(const ChainArgs& g, ChainState16<N>& st, int gq)
{
    constexpr ChainLayer L = N::layer(LI);
    const float* b = g.params + g.d.layer[L.desc].b_off;    // b_off is only dword aligned
    if constexpr ((L.desc == N::L_value && N::HV == 1) || (L.desc == N::L_mean && N::HM == 1)) {
        st.bias[0] = make_float4(b[0], 0.0f, 0.0f, 0.0f);
    } else if constexpr (L.desc == N::L_mean || L.desc == N::L_value) {
        const f32x4u v = *reinterpret_cast<const f32x4u*>(b);
        st.bias[0] = make_float4(v.x, v.y, v.z, v.w);
    } else {
#pragma unroll
        for (int a = 0; a < Chain16<N>::nout(LI); ++a) {
            const f32x4u v = *reinterpret_cast<const f32x4u*>(b + 16 * a + 4 * gq);
            st.bias[a] = make_float4(v.x, v.y, v.z, v.w);
        }
    }
}

template <class N, int LI>
__device__ __forceinline__ void chain16_epilogue(const ChainArgs& g, ChainState16<N>& st, int row, int gq, bool live)
{
    constexpr ChainLayer L = N::layer(LI);
    if constexpr (N::is_head(LI)) {                    // heads: mean (M,4) / value (M,1); lane group 0 holds them
        f32x4& y = st.t[2 * L.out0];
        const float4 bq = st.bias[0];
        y[0] += bq.x; y[1] += bq.y; y[2] += bq.z; y[3] += bq.w;
        if (live && gq == 0) {
            if constexpr (L.desc == N::L_mean && N::HM == 1) {
                if (g.io.mean) g.io.mean[row] = y[0];
            } else if constexpr (L.desc == N::L_mean) {
                if (g.io.mean) *reinterpret_cast<float4*>(g.io.mean + (size_t)row * 4) = make_float4(y[0], y[1], y[2], y[3]);
                if constexpr (N::HV != 4) {
                    if (g.rp_action) {     // k_reparam_fwd's arithmetic on the head still in registers (as chain_epilogue)
                        const float4 e = g.rp_eps[row];
                        float sd[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) sd[k] = g.rp_std_valid ? g.rp_std[k] : expf(g.rp_log_std[k]);
                        st.act = make_float4(tanhf(y[0] + sd[0] * e.x), tanhf(y[1] + sd[1] * e.y), tanhf(y[2] + sd[2] * e.z), tanhf(y[3] + sd[3] * e.w));
                        g.rp_action[row] = st.act;
                    }
                }
            } else if constexpr (N::HV == 4) {
                if (g.io.value) *reinterpret_cast<float4*>(g.io.value + (size_t)row * 4) = make_float4(y[0], y[1], y[2], y[3]);
                if (g.rp_action) {         // k_shac_head_fwd's arithmetic (as chain_epilogue)
                    const f32x4& mu = st.t[2 * N::t_mean];
                    const float4 e = g.rp_eps[row];
                    const float lo = g.rp_ls_lo, hi = g.rp_ls_hi;
                    st.act = make_float4(tanhf(mu[0] + e.x * expf(chain_clampf(y[0], lo, hi))), tanhf(mu[1] + e.y * expf(chain_clampf(y[1], lo, hi))),
                                         tanhf(mu[2] + e.z * expf(chain_clampf(y[2], lo, hi))), tanhf(mu[3] + e.w * expf(chain_clampf(y[3], lo, hi))));
                    g.rp_action[row] = st.act;
                }
            } else {
                if (g.io.value) g.io.value[row] = y[0];
            }
        }
    } else {
#pragma unroll
        for (int a = 0; a < Chain16<N>::nout(LI); ++a) {
            f32x4& y = st.t[2 * L.out0 + a];
            const float4 bq = st.bias[a];
            y[0] = act_fwd_c<L.relu>(y[0] + bq.x); y[1] = act_fwd_c<L.relu>(y[1] + bq.y);
            y[2] = act_fwd_c<L.relu>(y[2] + bq.z); y[3] = act_fwd_c<L.relu>(y[3] + bq.w);
        }
    }
}

// 16-row form of chain_pass_tile: lane (m, gq) holds features 4 gq + r of a 16-feature tile -> group gq = 0 holds the input's columns
template <class N>
__device__ __forceinline__ void chain16_pass_tile(const ChainArgs& g, ChainState16<N>& st, int row, int rc, int gq, bool live)
{
    if constexpr (N::PASS) {
        const int pw = g.d.in_dim[N::NB];
        const float* x = g.io.in[N::NB] + (size_t)rc * pw;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (gq == 0 && k < pw) ? x[k < pw ? k : pw - 1] : 0.0f;
        st.t[2 * N::t_pass] = f32x4{v[0], v[1], v[2], v[3]};
        st.t[2 * N::t_pass + 1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const vf_mlp_layer& D = g.d.layer[N::L_ident];
        if (D.save && live && gq == 0) {
            float* o = D.save + (size_t)row * D.save_ld + D.dst_col;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < pw) o[k] = v[k];
        }
    }
}

// saved copies of the previous layer's output, trickled out under this layer's items (see chain_store_setup / chain_deferred_store)
template <class N, int LI>
__device__ __forceinline__ void chain16_store_setup(const ChainArgs& g, ChainState16<N>& st, int rc, int gq)
{
    if constexpr (LI >= 1 && !N::is_head(LI >= 1 ? LI - 1 : 0)) {
        constexpr ChainLayer P = N::layer(LI - 1);
        const vf_mlp_layer& D = g.d.layer[P.desc];
        unsigned long b = D.save ? reinterpret_cast<unsigned long>(D.save + D.dst_col) : 0ul;
        asm volatile("" : "+s"(b));
        st.sv_base = (vf_gptr)b;
        st.sv_off = ((unsigned)rc * (unsigned)D.save_ld + 4u * gq) * 4u;        // bytes
    }
}

template <class N, int LI, int LOCAL>
__device__ __forceinline__ void chain16_deferred_store(const ChainState16<N>& st)
{
    if constexpr (LI >= 1 && !N::is_head(LI >= 1 ? LI - 1 : 0)) {
        using C = Chain16<N>;
        constexpr ChainLayer P = N::layer(LI - 1);
        constexpr int S = C::nout(LI - 1), per = (S + C::items(LI) - 1) / C::items(LI);
        constexpr int s0 = LOCAL * per, s1 = (LOCAL + 1) * per < S ? (LOCAL + 1) * per : S;
        if constexpr (s0 < s1) {
            if (st.sv_base) {
                const vf_gptr base = st.sv_base + st.sv_off;
#pragma unroll
                for (int a = s0; a < s1; ++a) {
                    const f32x4& y = st.t[2 * P.out0 + a];
                    *(vf_gfloat4*)(base + 16 * a * 4) = vf_st4{y[0], y[1], y[2], y[3]};
                }
            }
        }
    }
}

// Items (T, a) and (T, a + 1) -- same input tile, neighbouring output tiles -- run as ONE step with their MFMAs interleaved:
// v_mfma_f32_16x16x4_f32 issues every 32 cycles but a MFMA that accumulates into the result of the one before it waits 40
// (MI355X_MICROARCH.md, per-instruction constants), so four back-to-back MFMAs on one accumulator run at 80 % of the pipe; two
// accumulators alternating do not wait.  Every accumulator still sees its products in the same order: same bits.
template <class N, int I, bool SAVE = true>
__device__ __forceinline__ void chain16_items(const ChainArgs& g, ChainState16<N>& st, int lane, int row, bool live, int rc)
{
    using C = Chain16<N>;
    if constexpr (I < C::n_items()) {
        constexpr int li = C::layer_of(I), local = I - C::first_item(li);
        constexpr ChainLayer L = N::layer(li);
        constexpr int T = local / C::nout(li), a = local % C::nout(li);
        constexpr bool pair = (a % 2 == 0) && (a + 1 < C::nout(li));
        constexpr int step = pair ? 2 : 1;
        const int gq = lane >> 4;
        const float4 w = st.ring[I % kChain16Depth];
        if constexpr (I + kChain16Depth < C::n_items()) st.ring[I % kChain16Depth] = chain16_load<N, I + kChain16Depth>(g, lane);
        float4 w1 = w;
        if constexpr (pair) {
            w1 = st.ring[(I + 1) % kChain16Depth];
            if constexpr (I + 1 + kChain16Depth < C::n_items()) st.ring[(I + 1) % kChain16Depth] = chain16_load<N, I + 1 + kChain16Depth>(g, lane);
        }
        if constexpr (local == 0) chain16_bias_load<N, li>(g, st, gq);
        f32x4& acc = st.t[2 * L.out0 + a];
        f32x4& acc1 = st.t[2 * L.out0 + a + (pair ? 1 : 0)];
        if constexpr (T == 0) {
            acc = f32x4{0};
            if constexpr (pair) acc1 = f32x4{0};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float b;
            if constexpr (L.obs >= 0) b = st.x[L.obs][j];
            else b = st.t[2 * L.in0 + T][j];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(j == 0 ? w.x : j == 1 ? w.y : j == 2 ? w.z : w.w, b, acc, 0, 0, 0);
            if constexpr (pair) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(j == 0 ? w1.x : j == 1 ? w1.y : j == 2 ? w1.z : w1.w, b, acc1, 0, 0, 0);
        }
        if constexpr (SAVE && local == 0) chain16_store_setup<N, li>(g, st, rc, gq);
        if constexpr (SAVE) chain16_deferred_store<N, li, local>(st);
        if constexpr (SAVE && pair) chain16_deferred_store<N, li, local + 1>(st);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (local == 0) VF_TRACE(2 + 2 * li);
        if constexpr (local + step - 1 == C::items(li) - 1) {
            chain16_epilogue<N, li>(g, st, row, gq, live);
            VF_TRACE(3 + 2 * li);
        }
        chain16_items<N, I + step, SAVE>(g, st, lane, row, live, rc);
    }
}

template <class N, int I>
__device__ __forceinline__ void chain16_prologue(const ChainArgs& g, ChainState16<N>& st, int lane)
{
    if constexpr (I < kChain16Depth && I < Chain16<N>::n_items()) {
        st.ring[I] = chain16_load<N, I>(g, lane);
        chain16_prologue<N, I + 1>(g, st, lane);
    }
}

}  // namespace vf
