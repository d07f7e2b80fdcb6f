// vf_env_bwd_quad.hpp -- the reverse sweep over the sub-steps of one control interval with FOUR LANES PER AGENT (component layout,
// vf_quad.hpp), for the persistent reverse launch of vf_bptt_reverse.hip.
//
// In k_bptt_reverse lanes 16..63 used to replicate lane & 15: the adjoint of an interval is ~6.5 k VALU instructions of a lone wave
// (4.1 cycles each whatever the lane count), three quarters of every instruction recomputing what another lane already had
// (VERDICT r03).  In component layout the sub-step loop is 253 instead of ~810 instructions per sub-step.
#pragma once
#include "vf_quad.hpp"

#pragma clang fp contract(off)

namespace vf {

// derivs_bwd (vf_env_bwd_body.hpp) by components
__device__ __forceinline__ void derivs_bwd_c(const QuadLane& L, float q, float w, float ldq, float ldw, float& lq, float& lw, float& ltau)
{
    const float lr = 0.0f + q_mat3c(L.Jic, ldw);
    const float lc = -lr;
    const float Jw = q_mat3r(L.Jr, w);
    const float t0 = q_cross(Jw, lc);         // lam_w += (J w) x lc
    const float t1 = q_cross(lc, w);          // lam_(J w) = lc x w
    ltau = ltau + lr;
    lw = lw + t0;
    lw = lw + q_mat3c(L.Jc, t1);
    const float Lq = ldq * 0.5f, W = q_pure(L, w);
    lq = lq + qmul_c(L, Lq, q_conj(L, W));
    const float lW = qmul_c(L, q_conj(L, q), Lq);
    lw = lw + lW;                             // (vector lanes; lane 0 of an adjoint vector is never read)
}

// rotate_bwd: adjoint of r = imag(q (0, x) conj(q));  lx starts at 0 in the caller's scalar form (lx = 0 + lX)
__device__ __forceinline__ float rotate_bwd_c(const QuadLane& L, float q, float x, float lr, float& lq)
{
    const float X = q_pure(L, x), Lr = q_pure(L, lr);
    const float A = qmul_c(L, q, X);
    const float lA = qmul_c(L, Lr, q);
    const float lqc = qmul_c(L, q_conj(L, A), Lr);
    lq = lq + q_conj(L, lqc);
    lq = lq + qmul_c(L, lA, q_conj(L, X));
    return qmul_c(L, q_conj(L, q), lA);
}
// inv_rotate_bwd: adjoint of r = imag(conj(q) (0, x) q)
__device__ __forceinline__ float inv_rotate_bwd_c(const QuadLane& L, float q, float x, float lr, float& lq)
{
    const float X = q_pure(L, x), Lr = q_pure(L, lr);
    const float qc = q_conj(L, q);
    const float A = qmul_c(L, qc, X);
    const float lA = qmul_c(L, Lr, q_conj(L, q));
    lq = lq + qmul_c(L, q_conj(L, A), Lr);
    const float lqc = qmul_c(L, lA, q_conj(L, X));
    lq = lq + q_conj(L, lqc);
    return qmul_c(L, q, lA);
}

// The adjoint of the persistent state BETWEEN two steps of a persistent reverse sweep, one register per quantity (component layout):
// k_bptt_reverse loads it from the adjoint slab before the sweep and stores it after -- per step the slab round trip (8 granule
// stores, a fence, 6 + 1 dependent loads at the head of the next step) is gone.  ring: the delay ring's action adjoints (a float4
// per slot, lane k = component k); at most kRingRegs slots (else the sweep keeps the slab in the loop).
constexpr int kRingRegs = 4;
struct QuadCarry {
    float lp, lq, lv, lw, lwm, laa;
    float ring[kRingRegs];
};

// adjoint state carried through the sub-steps, one register each (component layout)
struct QuadAdj {
    float lq, lv, lw, lp;       // quaternion; vectors in lanes 1..3
    float lwm, lwd, lTd;        // rotor k
    float ldw_in;               // adjoint of the angular-acceleration state = dw of the LAST sub-step (vector)
};

// One sub-step of the reverse sweep (env_step_bwd_agent's loop body) by components.  q, v, w, wm0: the state at the head of the
// sub-step (v, w: lane 0 = 0); kl, kq: drag coefficients (vector lanes); wd, Td: rotor set-point / clamped thrust of lane k's rotor.
template <int INTEG, bool CTRL_DELAY>
__device__ __forceinline__ void substep_bwd_c(const vf_dyn_cfg& c, const QuadLane& L, float q, float v, float w, float wm0, float kl, float kq,
                                              float wd, float Td, float dt, float inv_m, QuadAdj& a)
{
    const float wm1 = CTRL_DELAY ? c.c_motor * wm0 + c.one_minus_c * wd : wm0;
    const float Tt = CTRL_DELAY ? (c.tm0 * (wm1 * wm1) + c.tm1 * wm1) + c.tm2 : Td;
    const float ft = q_mat4r(L.Br, Tt);                              // lane 0: F, lanes 1..3: tau
    const float vb = qmul_c(L, qmul_c(L, q_conj(L, q), q_pure(L, v)), q);
    const float F = qb<0>(ft);
    const float zf = L.k == 3 ? F : 0.0f;
    const float u = zf - (kl * vb + (kq * vb) * fabsf(vb));
    float lacc, ltau = 0.0f, lq_in;
    if constexpr (INTEG == VF_INT_EULER) {
        float dq, dw;
        derivs_c<false>(L, q, w, ft, dq, dw);
        const float qt = q + dq * dt;
        const float nn = sqrtf(q_sum4(qt * qt));
        const float rnn = 1.0f / nn;
        const float qn = qt * rnn;
        const float dotl = q_sum4(qn * a.lq);
        const float lqt = (a.lq - qn * dotl) * rnn;
        const float ldw = a.lw * dt + a.ldw_in;
        lacc = a.lv * dt;
        a.lv = a.lv + a.lp * dt;
        a.ldw_in = 0.0f;
        lq_in = lqt;
        derivs_bwd_c(L, q, w, lqt * dt, ldw, lq_in, a.lw, ltau);
    } else {
        const float ks[4] = {1.0f / 6.0f, 2.0f / 6.0f, 2.0f / 6.0f, 1.0f / 6.0f}, hs[4] = {0.0f, 0.5f, 0.5f, 1.0f};
        float qs[4], ws[4], dq = 0.0f, dw = 0.0f, sq = 0.0f;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            if (st == 0) {
                qs[0] = q;
                ws[0] = w;
            } else {
                const float h = hs[st] * dt;
                qs[st] = q + dq * h;
                ws[st] = w + dw * h;
            }
            derivs_c<true>(L, qs[st], ws[st], ft, dq, dw);
            sq = sq + dq * ks[st];
        }
        const float qt = q + sq * dt;
        const float nn = sqrtf(q_sum4(qt * qt));
        const float rnn = 1.0f / nn;
        const float qn = qt * rnn;
        const float dotl = q_sum4(qn * a.lq);
        const float lqt = (a.lq - qn * dotl) * rnn;
        const float lsw = a.lw * dt + a.ldw_in;
        a.ldw_in = 0.0f;
        lacc = a.lv * dt + a.lp * dt * (0.5f * dt);
        a.lv = a.lv + a.lp * dt;
        const float lsq = lqt * dt;
        lq_in = lqt;
        float ldq_c = lsq * ks[3], ldw_c = lsw * ks[3];
#pragma unroll
        for (int st = 3; st >= 0; --st) {
            float lqc = 0.0f, lwc = 0.0f;
            derivs_bwd_c(L, qs[st], ws[st], ldq_c, ldw_c, lqc, lwc, ltau);
            lq_in = lq_in + lqc;
            a.lw = a.lw + lwc;
            if (st > 0) {
                const float h = hs[st] * dt;
                ldq_c = lsq * ks[st - 1];
                ldq_c = ldq_c + lqc * h;
                ldw_c = lsw * ks[st - 1] + lwc * h;
            }
        }
    }
    // acc = rotate(q, u) / m + g
    const float lra = lacc * inv_m;
    const float lu = 0.0f + rotate_bwd_c(L, q, u, lra, lq_in);
    // u = z F - drag;  drag = kl vb + kq vb |vb|
    const float lvb = -lu * (kl + 2.0f * kq * fabsf(vb));
    a.lv = a.lv + inv_rotate_bwd_c(L, q, v, lvb, lq_in);
    // [F; tau] = B T
    const float lF = qb<3>(lu);
    const float lft = L.k == 0 ? lF : ltau;
    float lT = L.Bc[0] * qb<0>(lft) + L.Bc[1] * qb<1>(lft);
    lT = lT + L.Bc[2] * qb<2>(lft);
    lT = lT + L.Bc[3] * qb<3>(lft);
    if constexpr (CTRL_DELAY) {
        const float l1 = a.lwm + lT * (2.0f * c.tm0 * wm1 + c.tm1);
        a.lwd = a.lwd + c.one_minus_c * l1;
        a.lwm = c.c_motor * l1;
    } else {
        a.lTd = a.lTd + lT;
    }
    a.lq = lq_in;
}

}  // namespace vf
