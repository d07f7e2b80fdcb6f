// vf_dyn_quad.hpp -- the sub-steps of one control interval with FOUR LANES PER AGENT (component layout, vf_quad.hpp), for the
// persistent forward launch of vf_bptt_rollout.hip, whose waves hold 16 agents in 64 lanes.
//
// control_interval (vf_dyn_device.hpp) is ~300 VALU instructions per sub-step of a lone wave: 4 Hamilton products for the drag / thrust
// rotations, the rigid-body derivatives, 7 IEEE divisions and a square root.  In component layout: 4 x 10 for the products, one
// division per lane where the one-lane form has three or four, one fma per matrix column -- ~130.  The controller before the loop and
// the clamps after it are per agent, not per component, and stay replicated in the four lanes of a quad.  Same IEEE operations on the
// same values in the same order as the one-lane form: bit-identical (tests/test_bptt_gpu.py).
#pragma once
#include "vf_quad.hpp"

#pragma clang fp contract(off)

namespace vf {

struct QuadState {      // Agent's integrated fields, one register each: p v w acc aa vectors (lanes 1..3), q quaternion, wm / T rotor k
    float p, q, v, w, wm, T, acc, aa;
};

// One sub-step (control_interval's loop body: motor_substep, trans_substep, rot_substep) by components.
// wd: rotor k's pre-multiplied set-point (rotor_setpoint), Td: its clamped desired thrust; kl, kq, wind: vector lanes;
// zfac = (0, 0, 0, 1): the z selector of `zf = (k == 2 ? 1 : 0) * F`; gadd = (0, 0, 0, g_z)
template <int INTEG, bool CTRL_DELAY>
__device__ __forceinline__ void substep_c(const vf_dyn_cfg& c, const QuadLane& L, QuadState& x, float wd, float Td, float kl, float kq, float wind,
                                          float zfac, float gadd)
{
    const float dt = c.dt;
    // _run_motors + allocation
    if constexpr (CTRL_DELAY) {
        x.wm = c.c_motor * x.wm + wd;
        const float wp = x.wm + 0.0f;
        x.T = (c.tm0 * (wp * wp) + c.tm1 * x.wm) + c.tm2;
    } else {
        x.T = Td;
    }
    const float ft = q_mat4r(L.Br, x.T);                  // lane 0: F, lanes 1..3: tau
    // linear acceleration from the state at the start of the sub-step
    const float vq = q_pure(L, x.v + 0.0f);
    const float vb = qmul_c(L, qmul_c(L, q_conj(L, x.q), vq), x.q);
    const float drag = kl * vb + (kq * vb) * __builtin_fabsf(vb);
    const float zf = zfac * qb<0>(ft);
    const float u = zf - drag;
    const float ra = qmul_c(L, qmul_c(L, x.q, q_pure(L, u)), q_conj(L, x.q));
    const float acc = q_pure(L, ra / c.m + gadd);
    x.acc = acc;
    if constexpr (INTEG == VF_INT_EULER) {
        x.p = x.p + (x.v + wind) * dt;
        x.v = x.v + acc * dt;
        float dq, dw;
        derivs_c<true>(L, x.q, x.w, ft, dq, dw);
        x.q = x.q + dq * dt;
        x.w = x.w + dw * dt;
        x.aa = dw;
    } else {
        const float ks0 = 1.0f / 6.0f, ks1 = 2.0f / 6.0f;
        // translation (trans_substep): the stages see v + acc h dt
        float sp = 0.0f, sv = 0.0f;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const float h = st == 3 ? 1.0f : 0.5f;
            const float ks = (st == 0 || st == 3) ? ks0 : ks1;
            const float vc = st == 0 ? x.v : x.v + acc * h * dt;
            const float kp = (vc + wind) * ks, kv = acc * ks;
            sp = st == 0 ? kp : sp + kp;
            sv = st == 0 ? kv : sv + kv;
        }
        x.p = x.p + sp * dt;
        x.v = x.v + sv * dt;
        // rotation (rot_substep): tau frozen over the sub-step
        float qc = x.q, wc = x.w, sq = 0.0f, sw = 0.0f, dq = 0.0f, dw = 0.0f;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            if (st != 0) {
                const float h = st == 3 ? 1.0f : 0.5f;
                qc = x.q + dq * h * dt;
                wc = x.w + dw * h * dt;
            }
            derivs_c<true>(L, qc, wc, ft, dq, dw);
            const float ks = (st == 0 || st == 3) ? ks0 : ks1;
            const float kw = dw * ks;
            sw = st == 0 ? kw : sw + kw;
            const float kq4 = dq * ks;
            sq = st == 0 ? kq4 : sq + kq4;
        }
        x.q = x.q + sq * dt;
        x.w = x.w + sw * dt;
        x.aa = sw;
    }
    const float nn = vf_sqrt(q_sum4(x.q * x.q));
    x.q = x.q / nn;
}

struct NoCheckpointQuad {
    __device__ __forceinline__ void head_c(int, float, float, float, float) const {}
    __device__ __forceinline__ void end_c(float, float, float, float) const {}
};

// control_interval (vf_dyn_device.hpp) for an agent held by the four lanes of a quad: `s` is replicated in the quad on entry and on exit.
// CK: head_c(sub, q, v, w, wm) / end_c(p, q, v, w) see this lane's components (vectors: lane 0 = 0)
template <int ACT, int INTEG, bool CTRL_DELAY, class CK>
// L: quad_lane(c, lane), made once per launch by the caller (lane-dependent loads of the cfg matrices)
__device__ __forceinline__ void control_interval_quad(const vf_dyn_cfg& c, const QuadLane& L, Agent& s, const float* a, const float* kl, const float* kq,
                                                      bool vstrided, const CK& ck)
{
    float Td[4];
    desired_thrusts<ACT>(c, s, a, Td, vstrided);
    const int k = L.k;
    const float Td_c = q_sel4(k, Td[0], Td[1], Td[2], Td[3]);
    float wd_c = 0.0f;
    if constexpr (CTRL_DELAY) {      // rotor_setpoint for rotor k
        const float d3 = c.rot_tm1sq - c.rot_4tm0 * (c.tm2 - Td_c);
        wd_c = (c.one_minus_c) * (c.rot_scale * (c.rot_neg_tm1 + vf_sqrt(d3)));
    }
    QuadState x{q_sel3(k, s.p), q_sel4(k, s.q.w, s.q.x, s.q.y, s.q.z), q_sel3(k, s.v), q_sel3(k, s.w), q_sel4(k, s.wm[0], s.wm[1], s.wm[2], s.wm[3]),
                q_sel4(k, s.T[0], s.T[1], s.T[2], s.T[3]), q_sel3(k, s.acc), q_sel3(k, s.aa)};
    const float kl_c = q_sel3(k, kl), kq_c = q_sel3(k, kq), wind_c = q_sel3(k, s.wnd);
    const float zfac = k == 3 ? 1.0f : 0.0f, gadd = k == 3 ? c.g_z : 0.0f;
#pragma unroll 1
    for (int sub = 0; sub < c.interval_steps; ++sub) {
        ck.head_c(sub, x.q, x.v, x.w, x.wm);
        substep_c<INTEG, CTRL_DELAY>(c, L, x, wd_c, Td_c, kl_c, kq_c, wind_c, zfac, gadd);
    }
    ck.end_c(x.p, x.q, x.v, x.w);
    s.p[0] = qb<1>(x.p); s.p[1] = qb<2>(x.p); s.p[2] = qb<3>(x.p);
    s.q = Quat{qb<0>(x.q), qb<1>(x.q), qb<2>(x.q), qb<3>(x.q)};
    s.v[0] = qb<1>(x.v); s.v[1] = qb<2>(x.v); s.v[2] = qb<3>(x.v);
    s.w[0] = qb<1>(x.w); s.w[1] = qb<2>(x.w); s.w[2] = qb<3>(x.w);
    s.wm[0] = qb<0>(x.wm); s.wm[1] = qb<1>(x.wm); s.wm[2] = qb<2>(x.wm); s.wm[3] = qb<3>(x.wm);
    s.T[0] = qb<0>(x.T); s.T[1] = qb<1>(x.T); s.T[2] = qb<2>(x.T); s.T[3] = qb<3>(x.T);
    s.acc[0] = qb<1>(x.acc); s.acc[1] = qb<2>(x.acc); s.acc[2] = qb<3>(x.acc);
    s.aa[0] = qb<1>(x.aa); s.aa[1] = qb<2>(x.aa); s.aa[2] = qb<3>(x.aa);
    finish_interval(c, s);
}

}  // namespace vf
