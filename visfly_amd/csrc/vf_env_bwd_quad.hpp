// vf_env_bwd_quad.hpp -- the reverse sweep over the sub-steps of one control interval with FOUR LANES PER AGENT (component layout),
// for the persistent reverse launch of vf_bptt_reverse.hip, whose waves hold 16 agents in 64 lanes.
//
// In k_bptt_reverse lanes 16..63 used to replicate lane & 15: the adjoint of an interval is ~6.5 k VALU instructions of a lone wave
// (4.1 cycles each whatever the lane count), three quarters of every instruction recomputing what another lane already had
// (VERDICT r03).  Here the four lanes of a QUAD (lanes 4 m .. 4 m + 3 = agent slot m) hold the four COMPONENTS of the agent's
// quantities instead:
//     quaternions      lane k = component k of (w, x, y, z)
//     3-vectors        lanes 1 .. 3 = (x, y, z), lane 0 = 0 -- i.e. the pure quaternion (0, x) the rotations are written with
//     rotor quantities lane k = rotor k;    [F; tau] = B T: lane 0 = collective thrust, lanes 1 .. 3 = torque
// so that a Hamilton product is 10 instructions instead of 28 (4 products with a quad-broadcast operand -- DPP quad_perm, an operand
// modifier, no LDS --, 3 permuted + sign-flipped copies of the other operand, 3 adds), a 3x3 / 4x4 matrix product is one fma per
// column with the lane's ROW of the matrix in registers, a cross product 5, a dot product / norm 1 + 3.
// EVERY component is computed by the same sequence of IEEE operations on the same values as in the one-lane-per-agent form
// (env_step_bwd_agent's sub-step loop, which k_env_step_bwd keeps): term order of qmul (a.w, a.x, a.y, a.z), `a - b c` as
// `a + (-b) c`, the fma chains of mat3 / mat4, the association of every sum.  The two forms therefore agree to the bit
// (tests/test_bptt_gpu.py compares the persistent sweep with the launch-by-launch one bitwise), only WHERE a component lives differs.
#pragma once
#include "vf_env_device.hpp"

#pragma clang fp contract(off)

namespace vf {

#define VF_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))      // quad_perm: lane j of a quad reads lane a / b / c / d

template <int CTRL>
__device__ __forceinline__ float qdpp(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));   // (bound_ctrl: lets hipcc fold the move into the consuming VALU op)
}
template <int J>
__device__ __forceinline__ float qb(float x) { return qdpp<VF_QP(J, J, J, J)>(x); }                  // component J of the quad, in every lane
// NB: a cross-lane read must not sit INSIDE a lane-dependent conditional (`k == 3 ? qb<0>(x) : y` evaluates qb<0> only in the lanes
// with k == 3 -- C++ semantics, and DPP moves are convergent so hipcc keeps the branch: the source lane is then masked off and reads 0).
// Broadcast first, select afterwards.
__device__ __forceinline__ float q_nxt(float x) { return qdpp<VF_QP(0, 2, 3, 1)>(x); }               // vector lanes: x <- y, y <- z, z <- x
__device__ __forceinline__ float q_prv(float x) { return qdpp<VF_QP(0, 3, 1, 2)>(x); }               // vector lanes: x <- z, y <- x, z <- y
__device__ __forceinline__ float q_sx(float x, unsigned m) { return __uint_as_float(__float_as_uint(x) ^ m); }

struct QuadLane {
    int k;                               // lane & 3
    unsigned m1, m2, m3, mc;             // sign masks of qmul's terms 1..3 and of qconj for this lane
    float Jr[3], Jc[3], Jir[3], Jic[3];  // this lane's row / column of J and J^-1 (lanes 1..3 = rows 0..2; lane 0: zeros)
    float Br[4], Bc[4];                  // this lane's row / column of the allocation matrix B
};

__device__ __forceinline__ QuadLane quad_lane(const vf_dyn_cfg& c, int lane)
{
    QuadLane L;
    const int k = lane & 3;
    L.k = k;
    L.m1 = (k == 0 || k == 2) ? 0x80000000u : 0u;
    L.m2 = (k == 0 || k == 3) ? 0x80000000u : 0u;
    L.m3 = (k == 0 || k == 1) ? 0x80000000u : 0u;
    L.mc = k != 0 ? 0x80000000u : 0u;
    const int r = k > 0 ? k - 1 : 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float jr = c.J[3 * r + j], jc = c.J[3 * j + r], ir = c.Jinv[3 * r + j], ic = c.Jinv[3 * j + r];
        L.Jr[j] = k ? jr : 0.0f; L.Jc[j] = k ? jc : 0.0f; L.Jir[j] = k ? ir : 0.0f; L.Jic[j] = k ? ic : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { L.Br[j] = c.B[4 * k + j]; L.Bc[j] = c.B[4 * j + k]; }
    return L;
}

// this lane's component of a replicated quaternion / rotor quadruple, of a replicated 3-vector (lane 0: 0)
__device__ __forceinline__ float q_sel4(int k, float a, float b, float c2, float d)
{
    const float lo = (k & 1) ? b : a, hi = (k & 1) ? d : c2;
    return (k & 2) ? hi : lo;
}
__device__ __forceinline__ float q_sel3(int k, const float* v) { return q_sel4(k, 0.0f, v[0], v[1], v[2]); }
__device__ __forceinline__ float q_pure(const QuadLane& L, float x) { return L.k == 0 ? 0.0f : x; }       // Quat{0, x}
__device__ __forceinline__ float q_conj(const QuadLane& L, float x) { return q_sx(x, L.mc); }

// qmul (vf_dyn_device.hpp) by components: lane k gets ((a.w b[s0] +- a.x b[s1]) +- a.y b[s2]) +- a.z b[s3]
__device__ __forceinline__ float qmul_c(const QuadLane& L, float a, float b)
{
    float t = qb<0>(a) * b;
    t = t + qb<1>(a) * q_sx(qdpp<VF_QP(1, 0, 3, 2)>(b), L.m1);
    t = t + qb<2>(a) * q_sx(qdpp<VF_QP(2, 3, 0, 1)>(b), L.m2);
    t = t + qb<3>(a) * q_sx(qdpp<VF_QP(3, 2, 1, 0)>(b), L.m3);
    return t;
}
// ((t.w + t.x) + t.y) + t.z in every lane
__device__ __forceinline__ float q_sum4(float t)
{
    float s = qb<0>(t) + qb<1>(t);
    s = s + qb<2>(t);
    s = s + qb<3>(t);
    return s;
}
// mat3 (fma chain of a ROW with x): rows in lanes 1..3, x in lanes 1..3
__device__ __forceinline__ float q_mat3r(const float* row, float x)
{
    float acc = row[0] * qb<1>(x);
    acc = __builtin_fmaf(row[1], qb<2>(x), acc);
    acc = __builtin_fmaf(row[2], qb<3>(x), acc);
    return acc;
}
// mat3T_acc's sum: (A[j] x0 + A[3 + j] x1) + A[6 + j] x2 with the lane's COLUMN of A
__device__ __forceinline__ float q_mat3c(const float* col, float x)
{
    float s = col[0] * qb<1>(x);
    s = s + col[1] * qb<2>(x);
    s = s + col[2] * qb<3>(x);
    return s;
}
__device__ __forceinline__ float q_mat4r(const float* row, float x)
{
    float acc = row[0] * qb<0>(x);
    acc = __builtin_fmaf(row[1], qb<1>(x), acc);
    acc = __builtin_fmaf(row[2], qb<2>(x), acc);
    acc = __builtin_fmaf(row[3], qb<3>(x), acc);
    return acc;
}
// cross3 (vf_env_bwd_body.hpp): o = a[i + 1] b[i + 2] - a[i + 2] b[i + 1]
__device__ __forceinline__ float q_cross(float a, float b) { return q_nxt(a) * q_prv(b) - q_prv(a) * q_nxt(b); }

// derivs (vf_dyn_device.hpp) by components: dq = 0.5 q (0, w), dw = Jinv (tau - w x (J w)); tau in lanes 1..3
template <bool NEED_DW>
__device__ __forceinline__ void derivs_c(const QuadLane& L, float q, float w, float tau, float& dq, float& dw)
{
    const float p = qmul_c(L, q, q_pure(L, w));
    dq = p * 0.5f;
    if constexpr (NEED_DW) {
        const float Jw = q_mat3r(L.Jr, w);
        const float cr = __builtin_fmaf(q_nxt(w), q_prv(Jw), -(q_prv(w) * q_nxt(Jw)));     // torch.linalg.cross: fma(a_i, b_j, -(a_j b_i))
        dw = q_mat3r(L.Jir, tau - cr);
    }
}

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
