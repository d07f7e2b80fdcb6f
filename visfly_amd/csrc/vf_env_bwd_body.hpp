// vf_env_bwd_body.hpp -- the adjoint of one control interval + env epilogue for ONE agent as device code, shared by k_env_step_bwd
// (vf_env_bwd.hip) and the persistent reverse sweep of vf_bptt_reverse.hip.
#pragma once
#include "vf_env_device.hpp"
#include "vf_env_bwd_quad.hpp"
#include "vf_handles.hpp"

#pragma clang fp contract(off)

namespace vf {

struct BwdArgs {
    int N, G, g_drag, g_race;
    const float* tape;
    const float4* action;
    const float* d_obs;
    const float* d_reward;
    const unsigned char* done;
    float* adj;
    float4* d_action;
};

__device__ __forceinline__ Quat qscale(const Quat& a, float s) { return Quat{a.w * s, a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ void qacc(Quat& a, const Quat& b) { a.w += b.w; a.x += b.x; a.y += b.y; a.z += b.z; }
__device__ __forceinline__ void cross3(const float* a, const float* b, float* o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
// o += A^T x  (3x3 row-major)
__device__ __forceinline__ void mat3T_acc(const float* A, const float* x, float* o)
{
#pragma unroll
    for (int j = 0; j < 3; ++j) o[j] += A[j] * x[0] + A[3 + j] * x[1] + A[6 + j] * x[2];
}
__device__ __forceinline__ float in_closed(float x, float lo, float hi) { return (x >= lo && x <= hi) ? 1.0f : 0.0f; }

// adjoint of r = imag(conj(q) * (0,x) * q)  [inv_rotate]  (utils/maths.py:40-49)
__device__ __forceinline__ void inv_rotate_bwd(const Quat& q, const float* x, const float* lr, Quat& lq, float* lx)
{
    const Quat X{0.0f, x[0], x[1], x[2]}, L{0.0f, lr[0], lr[1], lr[2]};
    const Quat qc = qconj(q);
    const Quat A = qmul(qc, X);                 // r = A * q
    const Quat lA = qmul(L, qconj(q));
    qacc(lq, qmul(qconj(A), L));
    const Quat lqc = qmul(lA, qconj(X));        // A = conj(q) * X
    qacc(lq, qconj(lqc));
    const Quat lX = qmul(q, lA);                // conj(conj(q)) * lA
    lx[0] += lX.x; lx[1] += lX.y; lx[2] += lX.z;
}
// adjoint of r = imag(q * (0,x) * conj(q))  [rotate]  (utils/maths.py:32-38)
__device__ __forceinline__ void rotate_bwd(const Quat& q, const float* x, const float* lr, Quat& lq, float* lx)
{
    const Quat X{0.0f, x[0], x[1], x[2]}, L{0.0f, lr[0], lr[1], lr[2]};
    const Quat A = qmul(q, X);                  // r = A * conj(q)
    const Quat lA = qmul(L, q);                 // L * conj(conj(q))
    const Quat lqc = qmul(qconj(A), L);
    qacc(lq, qconj(lqc));
    qacc(lq, qmul(lA, qconj(X)));               // A = q * X
    const Quat lX = qmul(qconj(q), lA);
    lx[0] += lX.x; lx[1] += lX.y; lx[2] += lX.z;
}

constexpr int kSave = 14;  // per sub-step: q(4) v(3) w(3) wm(4)

// adjoint of (dq, dw) = derivs(q, w, tau)  (utils/maths.py:311,314): dq = 0.5 q (0,w), dw = Jinv (tau - w x (J w))
__device__ __forceinline__ void derivs_bwd(const vf_dyn_cfg& c, const Quat& q, const float* w, const Quat& ldq, const float* ldw,
                                           Quat& lq, float* lw, float* ltau)
{
    float lr[3] = {0, 0, 0};
    mat3T_acc(c.Jinv, ldw, lr);
    const float lc[3] = {-lr[0], -lr[1], -lr[2]};
    float Jw[3], t0[3], t1[3];
    mat3(c.J, w[0], w[1], w[2], Jw);
    cross3(Jw, lc, t0);        // lam_w += (Jw) x lc
    cross3(lc, w, t1);         // lam_(Jw) = lc x w
#pragma unroll
    for (int k = 0; k < 3; ++k) { ltau[k] += lr[k]; lw[k] += t0[k]; }
    mat3T_acc(c.J, t1, lw);
    const Quat L = qscale(ldq, 0.5f), W{0.0f, w[0], w[1], w[2]};
    qacc(lq, qmul(L, qconj(W)));
    const Quat lW = qmul(qconj(q), L);
    lw[0] += lW.x; lw[1] += lW.y; lw[2] += lW.z;
}

// Reverse pass of ONE control interval for agent i (see the kernel below for what it computes).  `lds_col` = this thread's column
// of the [S * kSave][STRIDE] parking area (STRIDE = threads per workgroup).
//
// CKPT: the forward kernel left the interval's sub-step tape (k_bptt_rollout's TapeCheckpoint) and the caller has brought this
// wave's record of the step into LDS (k_bptt_reverse: LDS-DMA one step ahead, double-buffered): `rec` = rows of 64 float4, row r =
// sub-step r (r < S), the pre-clamp end state (r = S), the step's inputs / outcome (S + 1), its drag granules (S + 2).  Rows 0 .. S are
// component-major -- the float4 at [k * 16 + m] holds component k of four quantities of agent slot m, vectors as (0, x, y, z):
// (q_k, v_k, w_k, rotor speed k) / (p_k, q_k, v_k, w_k) --, rows S + 1, S + 2 entry-major (entry k of slot m at [k * 16 + m]).  The
// replay of the interval -- a third of this function's instruction stream -- is skipped; same values as the replay computes (it
// runs the forward kernels' own sub-step functions), so the adjoint is bit-identical either way.  16 agents per wave.
//
// QUAD (with CKPT): four lanes per agent -- the wave's 16 agents sit in QUADS (lanes 4 m .. 4 m + 3 = agent slot m; the caller passes the
// same `i` to the four lanes of a quad) and the sub-step loop, two thirds of this function, runs in component layout
// (vf_env_bwd_quad.hpp).  Everything outside the loop is per agent, not per component, and stays replicated.  Bit-identical.
template <int KIND, int ACT, int INTEG, bool CTRL_DELAY, int STRIDE, bool CKPT = false, bool QUAD = false>
__device__ __forceinline__ void env_step_bwd_agent(const vf_dyn_cfg& c, const vf_env_cfg& e, const BwdArgs& g, int i, bool live, float* lds_col,
                                                   const float4* rec = nullptr, QuadCarry* carry = nullptr, const float* d_obs_lds = nullptr,
                                                   float* d_action_lds = nullptr, const QuadLane* quad = nullptr)
{
    // QUAD: `carry` holds the adjoint of the persistent state from step to step (g.adj is not touched), the observation gradient of
    // the wave's agents comes from d_obs_lds ([16 slots][16] floats, LDS; read iff g.d_obs != nullptr), the action gradient goes to
    // d_action_lds ([16 slots] float4, LDS; g.d_action is not written)
    static_assert(!QUAD || CKPT, "the component layout reads the sub-step tape");
    float* T = const_cast<float*>(g.tape);
    Agent s;
    Spares sp;
    const int rec_slot = QUAD ? (threadIdx.x >> 2) & 15 : threadIdx.x & 15;
    // ---- the step's inputs: pre-step body rates / angular acceleration / counters, the action this step consumed (oldest ring slot)
    // and the head it used, done / d_reward, drag coefficients.  CKPT: from rows S + 1, S + 2 of the record (k_bptt_rollout's
    // TapeCheckpoint::inputs / outcome / drag) -- in LDS since the previous step; else from the tape slab and the per-step rows
    int head = 0;
    float a[4];
    float kl[3], kq[3];
    bool done_in;
    float dr_in = 0.0f, gate_bits = 0.0f;
    if constexpr (CKPT) {
        const float4* rin = rec + (c.interval_steps + 1) * 64;
        float4 e0, e1, e2, e3;       // (reads the compiler does not order behind the LDS-DMA in flight: vf_quad.hpp)
        lds_read4_opaque<256>(rin + rec_slot, e0, e1, e2, e3);
        s.w[0] = e0.x; s.w[1] = e0.y; s.w[2] = e0.z; sp.vel = e0.w;
        s.aa[0] = e1.x; s.aa[1] = e1.y; s.aa[2] = e1.z; sp.omg = e1.w;
        a[0] = e2.x; a[1] = e2.y; a[2] = e2.z; a[3] = e2.w;
        done_in = e3.x != 0.0f;
        dr_in = e3.y;
        gate_bits = e3.z;
        if (c.delay_steps > 0) {
            head = __float_as_int(sp.vel);
            head = (unsigned)head < (unsigned)c.delay_steps ? head : 0;
        }
        if (g.g_drag >= 0) {
            float4 x, y;
            lds_read2_opaque(rin + 64 + rec_slot, x, y);
            kl[0] = x.y; kl[1] = x.z; kl[2] = x.w; kq[0] = y.y; kq[1] = y.z; kq[2] = y.w;
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) { kl[k] = c.k_lin[k]; kq[k] = c.k_quad[k]; }
        }
    } else {
        load_agent(T, g.G, i, s, sp);
        float4 an = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) an = g.action[i];
        if (c.delay_steps > 0) {
            head = __float_as_int(sp.vel);
            head = (unsigned)head < (unsigned)c.delay_steps ? head : 0;
            an = *granule(T, g.G, i, VF_G_RING + head);
        }
        a[0] = an.x; a[1] = an.y; a[2] = an.z; a[3] = an.w;
        if (g.g_drag >= 0) {
            const float4 x = *granule(T, g.G, i, g.g_drag), y = *granule(T, g.G, i, g.g_drag + 1);
            kl[0] = x.y; kl[1] = x.z; kl[2] = x.w; kq[0] = y.y; kq[1] = y.z; kq[2] = y.w;
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) { kl[k] = c.k_lin[k]; kq[k] = c.k_quad[k]; }
        }
        done_in = live && g.done != nullptr && g.done[i] != 0;
        if (live && g.d_reward) dr_in = g.d_reward[i];
        if constexpr (kind_is_racing(KIND)) {
            if (live) gate_bits = granule(T, g.G, i, g.g_race)->x;
        }
    }
    const float w0[3] = {s.w[0], s.w[1], s.w[2]}, al0[3] = {s.aa[0], s.aa[1], s.aa[2]};

    // ---- forward replay (same arithmetic as control_interval), parking sub-step inputs ----
    float Traw[4], Td[4], wd[4], disc[4];
    if constexpr (ACT == VF_ACT_BODYRATE) {
        const float Fc = (a[0] * c.acc_half + c.acc_mean) * c.m;
        float ev[3], t1[3], Jw[3], t3[3], cr[3], u[4];
#pragma unroll
        for (int k = 0; k < 3; ++k) ev[k] = (a[k + 1] * c.rate_half + c.rate_mean) - s.w[k];
        mat3(c.JP, ev[0], ev[1], ev[2], t1);
        mat3(c.J, s.w[0], s.w[1], s.w[2], Jw);
        cross3(s.w, Jw, cr);
        mat3(c.Dm, s.aa[0], s.aa[1], s.aa[2], t3);
        u[0] = Fc;
#pragma unroll
        for (int k = 0; k < 3; ++k) u[k + 1] = (t1[k] + cr[k]) - t3[k];
        mat4(c.Binv, u, Traw);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) Traw[k] = c.m * (a[k] * c.acc_half + c.acc_mean);
    }
    const int qk = threadIdx.x & 3;
    // QUAD: rotor qk's clamp / set-point only (one sqrt per lane instead of four)
    const float Traw_c = q_sel4(qk, Traw[0], Traw[1], Traw[2], Traw[3]);
    const float Td_c = clampf(Traw_c, c.T_min, c.T_max);
    const float disc_c = c.rot_tm1sq - c.rot_4tm0 * (c.tm2 - Td_c);
    const float wd_c = c.rot_scale * (c.rot_neg_tm1 + sqrtf(disc_c));
    if constexpr (!QUAD) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            Td[k] = clampf(Traw[k], c.T_min, c.T_max);
            disc[k] = c.rot_tm1sq - c.rot_4tm0 * (c.tm2 - Td[k]);
            wd[k] = c.rot_scale * (c.rot_neg_tm1 + sqrtf(disc[k]));
        }
    }
    const float dt = c.dt;
    const int S = c.interval_steps;
    const int mslot = rec_slot;
    if constexpr (CKPT) {
        // end row, component-major (TapeCheckpoint::end_c): the float4 at [k * 16 + slot] = (p_k, q_k, v_k, w_k), vectors as (0, x, y, z)
        float4 c0, c1, c2, c3;
        lds_read4_opaque<256>(rec + S * 64 + mslot, c0, c1, c2, c3);
        s.p[0] = c1.x; s.p[1] = c2.x; s.p[2] = c3.x;
        s.q = Quat{c0.y, c1.y, c2.y, c3.y};
        s.v[0] = c1.z; s.v[1] = c2.z; s.v[2] = c3.z;
        s.w[0] = c1.w; s.w[1] = c2.w; s.w[2] = c3.w;
    } else {
        // the forward kernels' own sub-step functions (control_interval's loop body): the replayed states ARE the forward's, bit for
        // bit (r03 restated the Euler sub-step here and normalised q with a reciprocal product where the forward divides: 1 ulp)
        float wdp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) wdp[k] = c.one_minus_c * wd[k];           // rotor_setpoint's pre-multiplied set-point
        for (int sub = 0; sub < S; ++sub) {
            float* sv = lds_col + (size_t)sub * kSave * STRIDE;
            sv[0 * STRIDE] = s.q.w; sv[1 * STRIDE] = s.q.x; sv[2 * STRIDE] = s.q.y; sv[3 * STRIDE] = s.q.z;
#pragma unroll
            for (int k = 0; k < 3; ++k) { sv[(4 + k) * STRIDE] = s.v[k]; sv[(7 + k) * STRIDE] = s.w[k]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) sv[(10 + k) * STRIDE] = s.wm[k];
            float ft[4];
            motor_substep<CTRL_DELAY>(c, Td, wdp, s.wm, s.T, ft);
            trans_substep<INTEG>(c, s.q, ft[0], kl, kq, c.wind, s.p, s.v, s.acc);
            rot_substep<INTEG>(c, ft + 1, s.q, s.w, s.aa);
        }
    }
    // pre-clamp values decide the clamp masks; clamped values are the step's outputs
    const float pm[3] = {in_closed(s.p[0], -c.pos_xy_lim, c.pos_xy_lim), in_closed(s.p[1], -c.pos_xy_lim, c.pos_xy_lim),
                         in_closed(s.p[2], c.pos_z_lo, c.pos_z_hi)};
    float vm[3], wmk[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        vm[k] = in_closed(s.v[k], -c.vel_lim, c.vel_lim);
        wmk[k] = in_closed(s.w[k], -c.omg_lim, c.omg_lim);
    }
    s.p[0] = clampf(s.p[0], -c.pos_xy_lim, c.pos_xy_lim);
    s.p[1] = clampf(s.p[1], -c.pos_xy_lim, c.pos_xy_lim);
    s.p[2] = clampf(s.p[2], c.pos_z_lo, c.pos_z_hi);
#pragma unroll
    for (int k = 0; k < 3; ++k) { s.v[k] = clampf(s.v[k], -c.vel_lim, c.vel_lim); s.w[k] = clampf(s.w[k], -c.omg_lim, c.omg_lim); }

    // ---- incoming adjoints of the post-step state (zero for agents reset at the end of this step) ----
    const bool cut = live ? done_in : true;
    float lp[3] = {0, 0, 0}, lv[3] = {0, 0, 0}, lw[3] = {0, 0, 0}, lwm[4] = {0, 0, 0, 0}, laa[3] = {0, 0, 0};
    Quat lq{0, 0, 0, 0};
    if constexpr (QUAD) {
        // (cross-lane reads first, selects afterwards: vf_env_bwd_quad.hpp)
        const float cp1 = qb<1>(carry->lp), cp2 = qb<2>(carry->lp), cp3 = qb<3>(carry->lp);
        const float cq0 = qb<0>(carry->lq), cq1 = qb<1>(carry->lq), cq2 = qb<2>(carry->lq), cq3 = qb<3>(carry->lq);
        const float cv1 = qb<1>(carry->lv), cv2 = qb<2>(carry->lv), cv3 = qb<3>(carry->lv);
        const float cw1 = qb<1>(carry->lw), cw2 = qb<2>(carry->lw), cw3 = qb<3>(carry->lw);
        const float cm0 = qb<0>(carry->lwm), cm1 = qb<1>(carry->lwm), cm2 = qb<2>(carry->lwm), cm3 = qb<3>(carry->lwm);
        const float ca1 = qb<1>(carry->laa), ca2 = qb<2>(carry->laa), ca3 = qb<3>(carry->laa);
        if (!cut) {
            lp[0] = cp1; lp[1] = cp2; lp[2] = cp3;
            lq = Quat{cq0, cq1, cq2, cq3};
            lv[0] = cv1; lv[1] = cv2; lv[2] = cv3;
            lw[0] = cw1; lw[1] = cw2; lw[2] = cw3;
            lwm[0] = cm0; lwm[1] = cm1; lwm[2] = cm2; lwm[3] = cm3;
            laa[0] = ca1; laa[1] = ca2; laa[2] = ca3;
            if (g.d_obs) {  // observation = [p, q, v + wind, w] of the post-step state (dynamics.py:779-786), in the env's obs_mode
                const float4* d4 = reinterpret_cast<const float4*>(d_obs_lds + 16 * rec_slot);
                float4 d0, d1, d2, d3;
                lds_read4_opaque<16>(d4, d0, d1, d2, d3);
                float dd[13] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w, d2.x, d2.y, d2.z, d2.w, d3.x};
                if constexpr (KIND == VF_ENV_RACING2) {
                    const float d16[16] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w, d2.x, d2.y, d2.z, d2.w, d3.x, d3.y, d3.z, d3.w};
                    race2_obs_bwd(e, d16, dd);
                }
                obs_variant_bwd(e, dd);
                lp[0] += dd[0]; lp[1] += dd[1]; lp[2] += dd[2];
                lq.w += dd[3]; lq.x += dd[4]; lq.y += dd[5]; lq.z += dd[6];
                lv[0] += dd[7]; lv[1] += dd[8]; lv[2] += dd[9];
                lw[0] += dd[10]; lw[1] += dd[11]; lw[2] += dd[12];
            }
        }
    } else if (!cut) {
        const float4 g0 = *granule(g.adj, g.G, i, VF_G_POS), g1 = *granule(g.adj, g.G, i, VF_G_QUAT);
        const float4 g2 = *granule(g.adj, g.G, i, VF_G_VEL), g3 = *granule(g.adj, g.G, i, VF_G_OMG);
        const float4 g4 = *granule(g.adj, g.G, i, VF_G_MOT), g6 = *granule(g.adj, g.G, i, VF_G_AACC);
        lp[0] = g0.y; lp[1] = g0.z; lp[2] = g0.w;
        lq = Quat{g1.x, g1.y, g1.z, g1.w};
        lv[0] = g2.y; lv[1] = g2.z; lv[2] = g2.w;
        lw[0] = g3.y; lw[1] = g3.z; lw[2] = g3.w;
        lwm[0] = g4.x; lwm[1] = g4.y; lwm[2] = g4.z; lwm[3] = g4.w;
        laa[0] = g6.y; laa[1] = g6.z; laa[2] = g6.w;
        if (g.d_obs) {  // observation = [p, q, v + wind, w] of the post-step state (dynamics.py:779-786), in the env's obs_mode
            float dd[13];
            if constexpr (KIND == VF_ENV_RACING2) {
                const float* d = g.d_obs + 16 * (size_t)i;
                float d16[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) d16[k] = d[k];
                race2_obs_bwd(e, d16, dd);
            } else {
                const float* d = g.d_obs + 13 * (size_t)i;
#pragma unroll
                for (int k = 0; k < 13; ++k) dd[k] = d[k];
            }
            obs_variant_bwd(e, dd);
            lp[0] += dd[0]; lp[1] += dd[1]; lp[2] += dd[2];
            lq.w += dd[3]; lq.x += dd[4]; lq.y += dd[5]; lq.z += dd[6];
            lv[0] += dd[7]; lv[1] += dd[8]; lv[2] += dd[9];
            lw[0] += dd[10]; lw[1] += dd[11]; lw[2] += dd[12];
        }
    }
    // reward gradient (computed on the pre-reset post-step state, so it survives a reset)
    if (live && g.d_reward && KIND == VF_ENV_NAV && e.reward_mode == VF_REWARD_NAV2) {
        // NavigationEnv2.get_reward (envs/NavigationEnv.py:185-224): 0.02 (v_along - |v_across|) toward the target - 0.001 |w| + success,
        // get_along_vertical_vector (:16-24): bn = base / (|base| + 1e-8), along = <v, bn>, across = v - bn along.  success is a constant.
        const float dr = dr_in;
        const float vv[3] = {s.v[0] + c.wind[0], s.v[1] + c.wind[1], s.v[2] + c.wind[2]};
        const float base[3] = {e.target[0] - s.p[0], e.target[1] - s.p[1], e.target[2] - s.p[2]};
        const float nb = norm3(base[0], base[1], base[2]), den = nb + 1e-8f;
        const float bn[3] = {base[0] / den, base[1] / den, base[2] / den};
        const float along = dot3(vv, bn);
        const float vert[3] = {vv[0] - bn[0] * along, vv[1] - bn[1] * along, vv[2] - bn[2] * along};
        const float away = norm3(vert[0], vert[1], vert[2]);
        const float g_away = dr * -0.02f;
        float lvert[3] = {0.f, 0.f, 0.f};
        if (away > 0.0f) { lvert[0] = g_away * vert[0] / away; lvert[1] = g_away * vert[1] / away; lvert[2] = g_away * vert[2] / away; }
        const float l_along = dr * 0.02f - dot3(lvert, bn);          // along enters directly and through across = v - bn along
        float lbn[3], lbase[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lv[k] += lvert[k] + l_along * bn[k];
            lbn[k] = l_along * vv[k] - lvert[k] * along;
        }
        const float proj = dot3(lbn, base);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lbase[k] = lbn[k] / den - (nb > 0.0f ? proj / (den * den) * base[k] / nb : 0.0f);
            lp[k] -= lbase[k];                                        // base = target - p
        }
        const float nw = norm3(s.w[0], s.w[1], s.w[2]);
        if (nw > 0.0f) {
#pragma unroll
            for (int k = 0; k < 3; ++k) lw[k] += dr * -0.001f * s.w[k] / nw;
        }
    } else if (live && g.d_reward && KIND == VF_ENV_NAV) {
        // NavigationEnv.get_reward (envs/NavigationEnv.py:84-99).  What autograd differentiates there: position, orientation,
        // velocity, angular velocity and -- through collision_vector = collision_point.detach() - position
        // (droneEnv.py:345-366) -- the distance / direction to the closest bbox face; success and the step counter are
        // constants.  clamp / clamp_max / clamp_min pass the gradient on the closed side, relu'(0) = 0, norm'(0) = 0.
        const float dr = dr_in;
        const float vv[3] = {s.v[0] + c.wind[0], s.v[1] + c.wind[1], s.v[2] + c.wind[2]};
        const Collision col = bbox_collision(e, s.p);
        const bool success = norm3(s.p[0] - e.target[0], s.p[1] - e.target[1], s.p[2] - e.target[2]) <= e.success_radius;
        const int step_count = __float_as_int(sp.omg) + 1;        // counter of THIS step (the tape holds the pre-step slab)
        float ltp[3] = {0.f, 0.f, 0.f}, lcv[3] = {0.f, 0.f, 0.f}, ldis = 0.0f;
        // t1 = clamp_max(<v, tp> / (1e-6 + |tp|), 10) * 0.01,  tp = target - p
        const float tp[3] = {e.target[0] - s.p[0], e.target[1] - s.p[1], e.target[2] - s.p[2]};
        const float ntp = norm3(tp[0], tp[1], tp[2]), den1 = 1e-6f + ntp, dot1 = dot3(vv, tp);
        if (dot1 / den1 <= 10.0f) {
            const float gq = dr * 0.01f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lv[k] += gq * tp[k] / den1;
                ltp[k] += gq * (vv[k] / den1 - (ntp > 0.0f ? dot1 / (den1 * den1) * tp[k] / ntp : 0.0f));
            }
        }
        // t2 = (clamp_min(acos(clamp(<dir, v> / (1e-6 + |v|), -1, 1)), pi/18) - pi/18) * -0.01,  dir = x_axis(q)
        const Quat& q = s.q;
        const float dir[3] = {1.0f - 2.0f * (q.y * q.y + q.z * q.z), 2.0f * (q.x * q.y + q.z * q.w), 2.0f * (q.x * q.z - q.y * q.w)};
        const float vn = norm3(vv[0], vv[1], vv[2]), den2 = 1e-6f + vn, dot2 = dot3(dir, vv);
        const float cs0 = dot2 / den2;
        const float thrd = (float)(3.14159265358979323846 / 18.0);
        if (cs0 >= -1.0f && cs0 <= 1.0f && (c.trig_mode == VF_TRIG_CR ? vfs_acosf_cr(cs0) : vfs_acosf_u10(cs0)) >= thrd) {
            const float gcs = dr * -0.01f * (-1.0f / sqrtf(1.0f - cs0 * cs0));
            float ldir[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                ldir[k] = gcs * vv[k] / den2;
                lv[k] += gcs * (dir[k] / den2 - (vn > 0.0f ? dot2 / (den2 * den2) * vv[k] / vn : 0.0f));
            }
            lq.w += ldir[1] * 2.0f * q.z - ldir[2] * 2.0f * q.y;
            lq.x += ldir[1] * 2.0f * q.y + ldir[2] * 2.0f * q.z;
            lq.y += ldir[0] * -4.0f * q.y + ldir[1] * 2.0f * q.x - ldir[2] * 2.0f * q.w;
            lq.z += ldir[0] * -4.0f * q.z + ldir[1] * 2.0f * q.w + ldir[2] * 2.0f * q.x;
        }
        // t3 .. t5: -1e-5 |q - 1|, -0.002 |v|, -0.002 |w|
        const float dqv[4] = {q.w - 1.0f, q.x, q.y, q.z};
        const float nq = norm4(dqv[0], dqv[1], dqv[2], dqv[3]), nw = norm3(s.w[0], s.w[1], s.w[2]);
        if (nq > 0.0f) {
            const float gq = dr * (float)-0.00001 / nq;
            lq.w += gq * dqv[0]; lq.x += gq * dqv[1]; lq.y += gq * dqv[2]; lq.z += gq * dqv[3];
        }
        // t8 = success * (max_steps - step) * 0.1 * (0.2 + 0.8 / (1 + |v|)): the only other |v| term
        const float sterm = (float)(success ? e.max_episode_steps - step_count : 0);
        const float lvn = dr * (-0.002f + sterm * 0.1f * 0.8f * (-1.0f / ((1.0f + vn) * (1.0f + vn))));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (vn > 0.0f) lv[k] += lvn * vv[k] / vn;
            if (nw > 0.0f) lw[k] += dr * -0.002f * s.w[k] / nw;
        }
        // t6 = -0.01 / (dis + 0.2);  t7 = relu(1 - dis) * relu(<cv, v> / (1e-6 + dis)) * -0.005
        ldis += dr * 0.01f / ((col.dis + 0.2f) * (col.dis + 0.2f));
        const float relu1 = 1.0f - col.dis > 0.0f ? 1.0f - col.dis : 0.0f;
        const float den7 = 1e-6f + col.dis, dot7 = dot3(col.vec, vv), ap = dot7 / den7;
        const float g7 = dr * -0.005f;
        if (1.0f - col.dis > 0.0f) ldis -= g7 * (ap > 0.0f ? ap : 0.0f);
        if (ap > 0.0f) {
            const float lap = g7 * relu1;
#pragma unroll
            for (int k = 0; k < 3; ++k) { lcv[k] += lap * vv[k] / den7; lv[k] += lap * col.vec[k] / den7; }
            ldis -= lap * dot7 / (den7 * den7);
        }
        if (col.dis > 0.0f) {
#pragma unroll
            for (int k = 0; k < 3; ++k) lcv[k] += ldis * col.vec[k] / col.dis;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) lp[k] -= ltp[k] + lcv[k];     // tp = target - p,  cv = const - p
    } else if (live && g.d_reward) {
        const float dr = dr_in;
        const float* tgt = e.target;
        if constexpr (kind_is_racing(KIND)) {
            int gate = __float_as_int(gate_bits);
            gate = (unsigned)gate < (unsigned)e.n_gates ? gate : 0;
            // CKPT (persistent sweep): a per-lane indexed load of the gate table would be a vector load queued behind the step's
            // HBM-cold mask / record fetches (vmcnt is one in-order counter); the table is 8 x 3 launch-uniform scalars: select
            float gsel[3];
            auto gate_pos = [&](int gi) {
                gsel[0] = e.gates[0][0]; gsel[1] = e.gates[0][1]; gsel[2] = e.gates[0][2];
#pragma unroll
                for (int q = 1; q < VF_MAX_GATES; ++q) {
                    const bool is = gi == q;
                    gsel[0] = is ? e.gates[q][0] : gsel[0]; gsel[1] = is ? e.gates[q][1] : gsel[1]; gsel[2] = is ? e.gates[q][2] : gsel[2];
                }
            };
            const float* gt = e.gates[gate];
            if constexpr (CKPT) { gate_pos(gate); gt = gsel; }
            const bool pass = norm3(s.p[0] - gt[0], s.p[1] - gt[1], s.p[2] - gt[2]) <= e.success_radius;
            gate = gate + (pass ? 1 : 0);
            gate = gate == e.n_gates ? 0 : gate;
            tgt = e.gates[gate];
            if constexpr (CKPT) { gate_pos(gate); tgt = gsel; }
        }
        const float c1 = (float)(-0.1 * 1 / 9), c2 = (float)-0.00001, c3 = (float)-0.002;
        const float dp[3] = {s.p[0] - tgt[0], s.p[1] - tgt[1], s.p[2] - tgt[2]};
        const float np_ = norm3(dp[0], dp[1], dp[2]);
        const float dqv[4] = {s.q.w - 1.0f, s.q.x, s.q.y, s.q.z};
        const float nq = norm4(dqv[0], dqv[1], dqv[2], dqv[3]);
        const float vv[3] = {s.v[0] + c.wind[0], s.v[1] + c.wind[1], s.v[2] + c.wind[2]};
        const float nv = norm3(vv[0], vv[1], vv[2]), nw = norm3(s.w[0], s.w[1], s.w[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (np_ > 0.0f) lp[k] += dr * c1 * dp[k] / np_;
            if (nv > 0.0f) lv[k] += dr * c3 * vv[k] / nv;
            if (nw > 0.0f) lw[k] += dr * c3 * s.w[k] / nw;
        }
        if (nq > 0.0f) {
            lq.w += dr * c2 * dqv[0] / nq; lq.x += dr * c2 * dqv[1] / nq;
            lq.y += dr * c2 * dqv[2] / nq; lq.z += dr * c2 * dqv[3] / nq;
        }
    }
    // _ugly_fix clamps
#pragma unroll
    for (int k = 0; k < 3; ++k) { lp[k] *= pm[k]; lv[k] *= vm[k]; lw[k] *= wmk[k]; }

    // ---- reverse sweep over the sub-steps ----
    float lwd[4] = {0, 0, 0, 0}, lTd[4] = {0, 0, 0, 0};
    float ldw_in[3] = {laa[0], laa[1], laa[2]};  // aa state = dw of the LAST sub-step
    const float inv_m = 1.0f / c.m;              // acc = rotate(q, u) / m: its adjoint multiplies by 1 / m (one division per step, see above)
    float lTraw_q = 0.0f;                        // QUAD: rotor qk's component of lTraw
    if constexpr (QUAD) {
        const QuadLane& QL = *quad;      // (the caller's, made once per launch: 20 lane-dependent loads of cfg matrices otherwise sit in every step)
        QuadAdj qa{q_sel4(qk, lq.w, lq.x, lq.y, lq.z), q_sel3(qk, lv), q_sel3(qk, lw), q_sel3(qk, lp),
                   q_sel4(qk, lwm[0], lwm[1], lwm[2], lwm[3]), 0.0f, 0.0f, q_sel3(qk, ldw_in)};
        const float kl_c = q_sel3(qk, kl), kq_c = q_sel3(qk, kq);
        // this lane's component of the sub-step records (component-major rows, TapeCheckpoint::head_c): ONE float4 per sub-step =
        // (q_k, v_k, w_k, rotor speed k) of agent slot mslot
        const float4* rq = rec + qk * 16 + mslot;
        for (int sub = S - 1; sub >= 0; --sub) {
            const float4 h = lds_read1_opaque(rq + sub * 64);
            substep_bwd_c<INTEG, CTRL_DELAY>(c, QL, h.x, h.y, h.z, h.w, kl_c, kq_c, wd_c, Td_c, dt, inv_m, qa);
        }
        lq = Quat{qb<0>(qa.lq), qb<1>(qa.lq), qb<2>(qa.lq), qb<3>(qa.lq)};
        lv[0] = qb<1>(qa.lv); lv[1] = qb<2>(qa.lv); lv[2] = qb<3>(qa.lv);
        lw[0] = qb<1>(qa.lw); lw[1] = qb<2>(qa.lw); lw[2] = qb<3>(qa.lw);
        lwm[0] = qb<0>(qa.lwm); lwm[1] = qb<1>(qa.lwm); lwm[2] = qb<2>(qa.lwm); lwm[3] = qb<3>(qa.lwm);
        float lTd_c = qa.lTd;
        if (CTRL_DELAY) lTd_c += qa.lwd / sqrtf(disc_c);
        lTraw_q = lTd_c * in_closed(Traw_c, c.T_min, c.T_max);
    }
    if constexpr (!QUAD)
    for (int sub = S - 1; sub >= 0; --sub) {
        Quat q;
        float v[3], w[3], wm0[4];
        if constexpr (CKPT) {      // (one lane per agent on the component-major rows: transposed reads)
            const float4 h0 = rec[sub * 64 + mslot], h1 = rec[sub * 64 + 16 + mslot], h2 = rec[sub * 64 + 32 + mslot], h3 = rec[sub * 64 + 48 + mslot];
            q = Quat{h0.x, h1.x, h2.x, h3.x};
            v[0] = h1.y; v[1] = h2.y; v[2] = h3.y;
            w[0] = h1.z; w[1] = h2.z; w[2] = h3.z;
            wm0[0] = h0.w; wm0[1] = h1.w; wm0[2] = h2.w; wm0[3] = h3.w;
        } else {
            const float* sv = lds_col + (size_t)sub * kSave * STRIDE;
            q = Quat{sv[0 * STRIDE], sv[1 * STRIDE], sv[2 * STRIDE], sv[3 * STRIDE]};
#pragma unroll
            for (int k = 0; k < 3; ++k) { v[k] = sv[(4 + k) * STRIDE]; w[k] = sv[(7 + k) * STRIDE]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) wm0[k] = sv[(10 + k) * STRIDE];
        }
        // recompute this sub-step's intermediates
        float wm1[4], Tt[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wm1[k] = CTRL_DELAY ? c.c_motor * wm0[k] + c.one_minus_c * wd[k] : wm0[k];
            Tt[k] = CTRL_DELAY ? (c.tm0 * (wm1[k] * wm1[k]) + c.tm1 * wm1[k]) + c.tm2 : Td[k];
        }
        float ft[4];
        mat4(c.B, Tt, ft);
        const Quat vq{0.0f, v[0], v[1], v[2]};
        const Quat vb = qmul(qmul(qconj(q), vq), q);
        const float vbv[3] = {vb.x, vb.y, vb.z};
        float u[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) u[k] = (k == 2 ? ft[0] : 0.0f) - (kl[k] * vbv[k] + (kq[k] * vbv[k]) * fabsf(vbv[k]));
        float lacc[3], ltau[3] = {0, 0, 0};
        Quat lq_in;
        if constexpr (INTEG == VF_INT_EULER) {
            float dq[4], dw[3];
            derivs(c, q, w, ft + 1, dq, dw);
            const Quat qt{q.w + dq[0] * dt, q.x + dq[1] * dt, q.y + dq[2] * dt, q.z + dq[3] * dt};
            const float nn = sqrtf(((qt.w * qt.w + qt.x * qt.x) + qt.y * qt.y) + qt.z * qt.z);
            // (the adjoint is held to 1e-5 of the reference's autograd, not to its bits: ONE IEEE division per normalisation, the
            // four quotients by |qt| are products with its reciprocal -- an IEEE fp32 division is 48 cycles of a lone wave's issue)
            const float rnn = 1.0f / nn;
            const Quat qn = qscale(qt, rnn);
            // normalise: q' = qt / |qt|
            const float dotl = qn.w * lq.w + qn.x * lq.x + qn.y * lq.y + qn.z * lq.z;
            const Quat lqt{(lq.w - qn.w * dotl) * rnn, (lq.x - qn.x * dotl) * rnn, (lq.y - qn.y * dotl) * rnn, (lq.z - qn.z * dotl) * rnn};
            // Euler update
            float ldw[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                ldw[k] = lw[k] * dt + ldw_in[k];
                lacc[k] = lv[k] * dt;
                lv[k] += lp[k] * dt;   // p' = p + (v + wind) dt ; lp, lw, lv carry through (identity part)
                ldw_in[k] = 0.0f;
            }
            lq_in = lqt;               // q~ = q + dq dt
            derivs_bwd(c, q, w, qscale(lqt, dt), ldw, lq_in, lw, ltau);
        } else {
            // RK4 over (q, w) with tau frozen: stage st sees q + dq_{st-1} h dt, w + dw_{st-1} h dt (h = .5, .5, 1);
            // q~ = q + dt sum ks dq_st, w' = w + dt sum ks dw_st, aa = sum ks dw_st
            const float ks[4] = {1.0f / 6.0f, 2.0f / 6.0f, 2.0f / 6.0f, 1.0f / 6.0f}, hs[4] = {0.0f, 0.5f, 0.5f, 1.0f};
            Quat qs[4];
            float ws[4][3], dq[4], dw[3], sq[4] = {0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                if (st == 0) {
                    qs[0] = q;
#pragma unroll
                    for (int k = 0; k < 3; ++k) ws[0][k] = w[k];
                } else {
                    const float h = hs[st] * dt;
                    qs[st] = Quat{q.w + dq[0] * h, q.x + dq[1] * h, q.y + dq[2] * h, q.z + dq[3] * h};
#pragma unroll
                    for (int k = 0; k < 3; ++k) ws[st][k] = w[k] + dw[k] * h;
                }
                derivs(c, qs[st], ws[st], ft + 1, dq, dw);
#pragma unroll
                for (int k = 0; k < 4; ++k) sq[k] += dq[k] * ks[st];
            }
            const Quat qt{q.w + sq[0] * dt, q.x + sq[1] * dt, q.y + sq[2] * dt, q.z + sq[3] * dt};
            const float nn = sqrtf(((qt.w * qt.w + qt.x * qt.x) + qt.y * qt.y) + qt.z * qt.z);
            const float rnn = 1.0f / nn;
            const Quat qn = qscale(qt, rnn);
            const float dotl = qn.w * lq.w + qn.x * lq.x + qn.y * lq.y + qn.z * lq.z;
            const Quat lqt{(lq.w - qn.w * dotl) * rnn, (lq.x - qn.x * dotl) * rnn, (lq.y - qn.y * dotl) * rnn, (lq.z - qn.z * dotl) * rnn};
            float lsw[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lsw[k] = lw[k] * dt + ldw_in[k];
                ldw_in[k] = 0.0f;
                // translation: p' = p + dt sum ks (v + acc h dt + wind), v' = v + dt sum ks acc  (sum ks = 1, sum ks h = 1/2)
                lacc[k] = lv[k] * dt + lp[k] * dt * (0.5f * dt);
                lv[k] += lp[k] * dt;
            }
            const Quat lsq = qscale(lqt, dt);
            lq_in = lqt;
            Quat ldq_c = qscale(lsq, ks[3]);                 // adjoint of the stage derivative currently being unwound
            float ldw_c[3] = {lsw[0] * ks[3], lsw[1] * ks[3], lsw[2] * ks[3]};
#pragma unroll
            for (int st = 3; st >= 0; --st) {
                Quat lqc{0, 0, 0, 0};
                float lwc[3] = {0, 0, 0};
                derivs_bwd(c, qs[st], ws[st], ldq_c, ldw_c, lqc, lwc, ltau);
                qacc(lq_in, lqc);                               // every stage state contains q and w once
#pragma unroll
                for (int k = 0; k < 3; ++k) lw[k] += lwc[k];
                if (st > 0) {                                   // ... and the previous stage's derivative times h dt
                    const float h = hs[st] * dt;
                    ldq_c = qscale(lsq, ks[st - 1]);
                    qacc(ldq_c, qscale(lqc, h));
#pragma unroll
                    for (int k = 0; k < 3; ++k) ldw_c[k] = lsw[k] * ks[st - 1] + lwc[k] * h;
                }
            }
        }
        // acc = rotate(q, u) / m + g
        float lra[3] = {lacc[0] * inv_m, lacc[1] * inv_m, lacc[2] * inv_m}, lu[3] = {0, 0, 0};
        rotate_bwd(q, u, lra, lq_in, lu);
        float lF = lu[2];
        // u = z F - drag ; drag = kl vb + kq vb |vb|
        float lvb[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) lvb[k] = -lu[k] * (kl[k] + 2.0f * kq[k] * fabsf(vbv[k]));
        inv_rotate_bwd(q, v, lvb, lq_in, lv);
        // [F; tau] = B T
        const float lft[4] = {lF, ltau[0], ltau[1], ltau[2]};
        float lT[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) lT[k] = c.B[k] * lft[0] + c.B[4 + k] * lft[1] + c.B[8 + k] * lft[2] + c.B[12 + k] * lft[3];
        if constexpr (CTRL_DELAY) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float l1 = lwm[k] + lT[k] * (2.0f * c.tm0 * wm1[k] + c.tm1);
                lwd[k] += c.one_minus_c * l1;
                lwm[k] = c.c_motor * l1;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) lTd[k] += lT[k];
        }
        lq = lq_in;
    }
    // ---- rotor set-point, clamp, controller, de-normalisation ----
    float lTraw[4];
    if constexpr (QUAD) {
        lTraw[0] = qb<0>(lTraw_q); lTraw[1] = qb<1>(lTraw_q); lTraw[2] = qb<2>(lTraw_q); lTraw[3] = qb<3>(lTraw_q);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (CTRL_DELAY) lTd[k] += lwd[k] / sqrtf(disc[k]);  // d wd / d Td = 1 / sqrt(tm1^2 - 4 tm0 (tm2 - Td))
            lTraw[k] = lTd[k] * in_closed(Traw[k], c.T_min, c.T_max);
        }
    }
    float la[4];
    if constexpr (ACT == VF_ACT_BODYRATE) {
        float lu4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            lu4[k] = c.Binv[k] * lTraw[0] + c.Binv[4 + k] * lTraw[1] + c.Binv[8 + k] * lTraw[2] + c.Binv[12 + k] * lTraw[3];
        la[0] = lu4[0] * c.m * c.acc_half;
        const float ltd[3] = {lu4[1], lu4[2], lu4[3]};
        float le[3] = {0, 0, 0};
        mat3T_acc(c.JP, ltd, le);
        float Jw[3], t0[3], t1[3];
        mat3(c.J, w0[0], w0[1], w0[2], Jw);
        cross3(Jw, ltd, t0);
        cross3(ltd, w0, t1);
        float nd[3] = {-ltd[0], -ltd[1], -ltd[2]};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            la[k + 1] = le[k] * c.rate_half;
            lw[k] += t0[k] - le[k];
        }
        mat3T_acc(c.J, t1, lw);
        float laa0[3] = {0, 0, 0};
        mat3T_acc(c.Dm, nd, laa0);
        laa[0] = laa0[0]; laa[1] = laa0[1]; laa[2] = laa0[2];
        (void)al0;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) la[k] = lTraw[k] * c.m * c.acc_half;
        laa[0] = laa[1] = laa[2] = 0.0f;
    }

    // ---- write the adjoint of the pre-step state; hand the action gradient out ----
    if constexpr (QUAD) {
        carry->lp = q_sel3(qk, lp);
        carry->lq = q_sel4(qk, lq.w, lq.x, lq.y, lq.z);
        carry->lv = q_sel3(qk, lv);
        carry->lw = q_sel3(qk, lw);
        carry->lwm = q_sel4(qk, lwm[0], lwm[1], lwm[2], lwm[3]);
        carry->laa = q_sel3(qk, laa);
        float dact_c = q_sel4(qk, la[0], la[1], la[2], la[3]);
        if (c.delay_steps > 0) {       // the ring slot logic of the slab form below, on registers (head is launch-uniform)
#pragma unroll
            for (int q = 0; q < kRingRegs; ++q) {
                if (q == head) {
                    const float pushed = cut ? 0.0f : carry->ring[q];
                    carry->ring[q] = dact_c;
                    dact_c = pushed;
                } else if (cut) {
                    carry->ring[q] = 0.0f;
                }
            }
        }
        d_action_lds[rec_slot * 4 + qk] = dact_c;
        return;
    }
    float4 dact = make_float4(la[0], la[1], la[2], la[3]);
    if (c.delay_steps > 0) {
        // the consumed action sat in ring[head]; the action passed to this step was pushed into the same
        // slot, so ITS gradient is whatever later steps accumulated on that slot
        float4* slot = granule(g.adj, g.G, i, VF_G_RING + head);
        const float4 pushed = cut ? make_float4(0.f, 0.f, 0.f, 0.f) : *slot;
        *slot = dact;
        dact = pushed;
        if (cut)
            for (int q = 0; q < c.delay_steps; ++q)
                if (q != head) *granule(g.adj, g.G, i, VF_G_RING + q) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    *granule(g.adj, g.G, i, VF_G_POS) = make_float4(0.f, lp[0], lp[1], lp[2]);
    *granule(g.adj, g.G, i, VF_G_QUAT) = make_float4(lq.w, lq.x, lq.y, lq.z);
    *granule(g.adj, g.G, i, VF_G_VEL) = make_float4(0.f, lv[0], lv[1], lv[2]);
    *granule(g.adj, g.G, i, VF_G_OMG) = make_float4(0.f, lw[0], lw[1], lw[2]);
    *granule(g.adj, g.G, i, VF_G_MOT) = make_float4(lwm[0], lwm[1], lwm[2], lwm[3]);
    *granule(g.adj, g.G, i, VF_G_THR) = make_float4(0.f, 0.f, 0.f, 0.f);
    *granule(g.adj, g.G, i, VF_G_AACC) = make_float4(0.f, laa[0], laa[1], laa[2]);
    *granule(g.adj, g.G, i, VF_G_ACC) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) g.d_action[i] = dact;
}

}  // namespace vf
