// Per-agent rigid-body arithmetic for the fused control-interval kernels (gfx950).
//
// One thread owns one agent; the whole state lives in VGPRs across all sub-steps.
// The arithmetic reproduces the reference's fp32 rounding sequence exactly
// (SURVEY.md App. A / B.4): every elementwise op is rounded separately (this file is
// compiled with -ffp-contract=off), and fused multiply-adds appear only where the
// reference's BLAS matmuls (k-ordered FMA chains, K in {3,4}) and torch.linalg.cross
// fuse.  Division and sqrt are the IEEE correctly rounded forms (hipcc default).
//
// Reference: envs/base/dynamics.py:319-382,389-413,505-554,692-714; utils/maths.py:168-174,
// 226-233,300-351.
#pragma once
#include <hip/hip_runtime.h>

#include "vf_common.hpp"
#include "vf_xmath.hpp"
#include "visfly_amd.h"

#pragma clang fp contract(off)

#ifndef VF_SUBSTEP_UNROLL
#define VF_SUBSTEP_UNROLL 1
#endif
// 1: IEEE-exact fast paths for sqrt and for the division by the (launch-uniform) mass, each behind a wave-uniform range
// guard; every other value of the guard takes the compiler's own expansion.  A/B knob of tools/env_step_probe.hip, which
// also checks both fast paths exhaustively against `sqrtf` / `x / m`.
#ifndef VF_FAST_EXACT
#define VF_FAST_EXACT 0
#endif

namespace vf {

struct Quat {
    float w, x, y, z;
};

struct Agent {
    float p[3];
    Quat q;
    float v[3];
    float w[3];
    float wm[4];  // motor omega
    float T[4];   // rotor thrusts
    float aa[3];  // angular acceleration of the last sub-step
    float acc[3]; // linear acceleration of the last sub-step
    float t;
    float wnd[3]; // wind velocity of this control interval: vf_dyn_cfg.wind, or the agent's row of vf_*_set_wind (not slab state)
};

// Hamilton product; term order and rounding of utils/maths.py:168-174
__device__ __forceinline__ Quat qmul(const Quat& a, const Quat& b)
{
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return r;
}

__device__ __forceinline__ Quat qconj(const Quat& a) { return Quat{a.w, -a.x, -a.y, -a.z}; }

// ---- exact fp32 sqrt / division by a launch-uniform constant without the compiler's range handling ----------------
// hipcc expands sqrtf (denormals preserved) to: rescale tiny inputs by 2^32, v_sqrt_f32 (<= 1 ulp), pick among the
// result and its two neighbours by the sign of the fma residuals, undo the rescale, patch 0 / inf -- 16 instructions, 88
// cycles for a lone wave (tools/valu_cost_probe).  For x = 0 or x >= 2^-96 (finite or not) the rescale and the patch are
// no-ops: sqrt_exact_core is the remaining 9 instructions and returns the same bits.
__device__ __forceinline__ float sqrt_exact_core(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __int_as_float(__float_as_int(s) - 1), su = __int_as_float(__float_as_int(s) + 1);
    const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
    s = rd <= 0.0f ? sd : s;
    s = ru > 0.0f ? su : s;
    return s;
}
__device__ __forceinline__ float vf_sqrt(float x)
{
#if VF_FAST_EXACT == 2
    return sqrt_exact_core(x);                          // measurement only: no guard
#elif VF_FAST_EXACT
    if (__builtin_amdgcn_ballot_w64(x < 0x1p-96f && x != 0.0f) == 0) return sqrt_exact_core(x);   // wave-uniform
#endif
    return sqrtf(x);
}
// x / m for a constant m with y = 1 / m (IEEE): twice-corrected reciprocal product.  Equal to the IEEE quotient for every
// x with |x| in [2^-100, 2^100] (the residuals are exact there; tools/div_const_probe.hip, tools/env_step_probe.hip check all
// 2^32 numerators); for x = 0 it returns +0 / -0 with possibly the other sign.
__device__ __forceinline__ float div_const_core(float x, float m, float y)
{
    float q = x * y;
    float e = __builtin_fmaf(-m, q, x);
    q = __builtin_fmaf(e, y, q);
    e = __builtin_fmaf(-m, q, x);
    return __builtin_fmaf(e, y, q);
}

// th.clamp: min(max(v, lo), hi) with NaN passing through
__device__ __forceinline__ float clampf(float v, float lo, float hi)
{
    float r = v < lo ? lo : v;
    return r > hi ? hi : r;
}

// (3x3) @ x as the k-ordered FMA chain of the reference's sgemm
__device__ __forceinline__ void mat3(const float* __restrict__ A, float x0, float x1, float x2, float* o)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float acc = A[3 * i] * x0;
        acc = __builtin_fmaf(A[3 * i + 1], x1, acc);
        acc = __builtin_fmaf(A[3 * i + 2], x2, acc);
        o[i] = acc;
    }
}

__device__ __forceinline__ void mat4(const float* __restrict__ A, const float* x, float* o)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float acc = A[4 * i] * x[0];
        acc = __builtin_fmaf(A[4 * i + 1], x[1], acc);
        acc = __builtin_fmaf(A[4 * i + 2], x[2], acc);
        acc = __builtin_fmaf(A[4 * i + 3], x[3], acc);
        o[i] = acc;
    }
}

// Integrator._get_derivatives for (q, omega)  (utils/maths.py:311,314)
__device__ __forceinline__ void derivs(const vf_dyn_cfg& c, const Quat& q, const float* w, const float* tq,
                                       float* dq, float* dw)
{
    const Quat wq{0.0f, w[0], w[1], w[2]};
    const Quat p = qmul(q, wq);
    dq[0] = p.w * 0.5f;
    dq[1] = p.x * 0.5f;
    dq[2] = p.y * 0.5f;
    dq[3] = p.z * 0.5f;
    float Jw[3];
    mat3(c.J, w[0], w[1], w[2], Jw);
    // torch.linalg.cross contracts to fma(a_i, b_j, -(a_j*b_i))
    const float c0 = __builtin_fmaf(w[1], Jw[2], -(w[2] * Jw[1]));
    const float c1 = __builtin_fmaf(w[2], Jw[0], -(w[0] * Jw[2]));
    const float c2 = __builtin_fmaf(w[0], Jw[1], -(w[1] * Jw[0]));
    mat3(c.Jinv, tq[0] - c0, tq[1] - c1, tq[2] - c2, dw);
}

// cross() helper of utils/maths.py:392-394: separately rounded products, then "+ 0"
__device__ __forceinline__ void cross_helper(const float* a, const float* b, float* o)
{
    o[0] = (a[1] * b[2] - a[2] * b[1]) + 0.0f;
    o[1] = (a[2] * b[0] - a[0] * b[2]) + 0.0f;
    o[2] = (a[0] * b[1] - a[1] * b[0]) + 0.0f;
}

// Geometric SO(3) controller of the velocity / position action types (dynamics.py:414-452 / :453-496),
// evaluated once per control interval; the reference walks the agents in a Python loop (:446-450).
// x.norm(dim=0) = FMA chain + IEEE sqrt, 3x3 products = k-ordered FMA chains.
// atan2 is SLEEF's atan2f_u10 restated (what torch.atan2 runs); sin / cos are SLEEF's u10 routines restated (torch runs
// closed-source MKL VML there; one ulp apart for ~2 % of the arguments), see vf_xmath.hpp: bit-identical to the CPU oracle,
// and held to a stated tolerance against the reference for these two action types (tests/test_dyn_gpu.py).
template <bool POSITION>
__device__ __forceinline__ void geometric_controller(const vf_dyn_cfg& c, const Agent& s, const float* a, float* Td, bool strided_v)
{
    float cmd[4];   // _de_normalize :716-730 -> [yaw, x, y, z]
    cmd[0] = a[0] * c.yaw_half + c.yaw_mean;
#pragma unroll
    for (int k = 1; k < 4; ++k) cmd[k] = a[k] * c.vel_half + c.vel_mean;
    float F[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float a_des;
        if constexpr (POSITION) {
            const float v_des = c.pos_d * (cmd[k + 1] - s.p[k]);   // :456
            a_des = c.vel_d * (v_des - s.v[k]);                    // :457
        } else {
            a_des = c.vel_p * (cmd[k + 1] - s.v[k]);               // :416
        }
        F[k] = c.m * (a_des - (k == 2 ? c.g_z : 0.0f));            // :417,458
    }
    const Quat& q = s.q;
    const float yaw_cur = vfs_atan2f_u10(2.0f * (q.w * q.z + q.x * q.y), 1.0f - 2.0f * (q.y * q.y + q.z * q.z));   // maths.py:248
    float yaw_des, gain;
    if constexpr (POSITION) {
        yaw_des = cmd[0];                                          // :461
        gain = c.pos_d;                                            // :468
    } else {
        const float vn = sqrtf(__builtin_fmaf(s.v[1], s.v[1], s.v[0] * s.v[0]));   // :421
        // :423-427.  torch.atan2 = SLEEF for contiguous operands, glibc's atan2f for strided ones; the velocity rows are strided
        // whenever the last full reset was given velocities (vf_xmath.hpp, vfs_atan2f_glibc; DynArgs.vstrided), wave-uniform
        yaw_des = vn > 0.1f ? (strided_v ? vfs_atan2f_glibc(s.v[1], s.v[0]) : vfs_atan2f_u10(s.v[1], s.v[0])) : yaw_cur;
        gain = c.vel_d;                                            // :433
    }
    float ye = yaw_des - yaw_cur;
    const bool crm = c.trig_mode == VF_TRIG_CR;       // wave-uniform: fp64-evaluated sin / cos rounded once, or SLEEF u10
    ye = crm ? vfs_atan2f_u10(vfs_sinf_cr(ye), vfs_cosf_cr(ye)) : vfs_atan2f_u10(vfs_sinf_u10(ye), vfs_cosf_u10(ye));   // :432,467
    const float yaw_spd = ye * gain * 2.0f;
    // gross thrust = (conj(q) * (0, F) * q).imag[2]               :435, maths.py:49,103
    const Quat fb = qmul(qmul(Quat{q.w, -q.x, -q.y, -q.z}, Quat{0.0f, F[0], F[1], F[2]}), q);
    float R[3][3];   // Quaternion.R (maths.py:116-120)
    R[0][0] = 1.0f - 2.0f * (q.y * q.y + q.z * q.z); R[0][1] = 2.0f * (q.x * q.y - q.z * q.w); R[0][2] = 2.0f * (q.x * q.z + q.y * q.w);
    R[1][0] = 2.0f * (q.x * q.y + q.z * q.w); R[1][1] = 1.0f - 2.0f * (q.x * q.x + q.z * q.z); R[1][2] = 2.0f * (q.y * q.z - q.x * q.w);
    R[2][0] = 2.0f * (q.x * q.z - q.y * q.w); R[2][1] = 2.0f * (q.y * q.z + q.x * q.w); R[2][2] = 1.0f - 2.0f * (q.x * q.x + q.y * q.y);
    // desired frame :437-442
    const float fn = sqrtf(__builtin_fmaf(F[2], F[2], __builtin_fmaf(F[1], F[1], F[0] * F[0])));
    const float b3[3] = {F[0] / fn, F[1] / fn, F[2] / fn};
    const float c1[3] = {crm ? vfs_cosf_cr(yaw_des) : vfs_cosf_u10(yaw_des), crm ? vfs_sinf_cr(yaw_des) : vfs_sinf_u10(yaw_des), 0.0f};
    float b2[3], b1[3];
    cross_helper(b3, c1, b2);
    const float bn = sqrtf(__builtin_fmaf(b2[2], b2[2], __builtin_fmaf(b2[1], b2[1], b2[0] * b2[0])));
#pragma unroll
    for (int k = 0; k < 3; ++k) b2[k] = b2[k] / bn;
    cross_helper(b2, b3, b1);
    float Rd[3][3];   // columns b1, b2, b3
#pragma unroll
    for (int r = 0; r < 3; ++r) { Rd[r][0] = b1[r]; Rd[r][1] = b2[r]; Rd[r][2] = b3[r]; }
    // :446-450 per agent: A = Rd^T R, Bm = R^T Rd (= A^T to the bit: products commute, same k order)
    float A[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            A[i][j] = __builtin_fmaf(Rd[2][i], R[2][j], __builtin_fmaf(Rd[1][i], R[1][j], Rd[0][i] * R[0][j]));
    const float m12 = 0.5f * (A[1][2] - A[2][1]), m02 = 0.5f * (A[0][2] - A[2][0]), m01 = 0.5f * (A[0][1] - A[1][0]);
    const float pose[3] = {m12, -m02, m01};
    float ang[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        ang[i] = __builtin_fmaf(A[i][2], yaw_spd, __builtin_fmaf(A[i][1], 0.0f, A[i][0] * 0.0f)) - s.w[i];
    float t1[3], t2[3], inner[3], tau[3], cr[3];
    mat3(c.Pm, pose[0], pose[1], pose[2], t1);
    if constexpr (POSITION) {   // :489-494
        float t3[3], Jw[3];
        mat3(c.P12, ang[0], ang[1], ang[2], t2);
        mat3(c.Dm, s.aa[0], s.aa[1], s.aa[2], t3);
        mat3(c.J, s.w[0], s.w[1], s.w[2], Jw);
        cross_helper(s.w, Jw, cr);
#pragma unroll
        for (int k = 0; k < 3; ++k) inner[k] = ((t1[k] + t2[k]) - t3[k]) - cr[k];
    } else {                    // :451
        mat3(c.Pm, ang[0], ang[1], ang[2], t2);
        cross_helper(s.w, s.w, cr);
#pragma unroll
        for (int k = 0; k < 3; ++k) inner[k] = (t1[k] + t2[k]) - cr[k];
    }
    mat3(c.J, inner[0], inner[1], inner[2], tau);
    const float u[4] = {fb.z, tau[0], tau[1], tau[2]};
    mat4(c.Binv, u, Td);        // :453,496
}

// De-normalise the (delayed) action and run the low-level controller once per control
// interval -> clamped desired rotor thrusts (dynamics.py:692-714,389-413,501).
template <int ACT>
__device__ __forceinline__ void desired_thrusts(const vf_dyn_cfg& c, const Agent& s, const float* a, float* Td, bool vstrided = false)
{
    if constexpr (ACT == VF_ACT_BODYRATE) {
        const float Fc = (a[0] * c.acc_half + c.acc_mean) * c.m;
        float e[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) e[k] = (a[k + 1] * c.rate_half + c.rate_mean) - s.w[k];
        float t1[3], Jw[3], t3[3];
        mat3(c.JP, e[0], e[1], e[2], t1);
        const float w0 = s.w[0] + 0.0f, w1 = s.w[1] + 0.0f, w2 = s.w[2] + 0.0f;
        mat3(c.J, w0, w1, w2, Jw);
        // cross() helper of utils/maths.py:392-394: separately rounded, then "+ 0"
        float cr[3];
        cr[0] = (w1 * Jw[2] - w2 * Jw[1]) + 0.0f;
        cr[1] = (w2 * Jw[0] - w0 * Jw[2]) + 0.0f;
        cr[2] = (w0 * Jw[1] - w1 * Jw[0]) + 0.0f;
        mat3(c.Dm, s.aa[0], s.aa[1], s.aa[2], t3);
        float u[4];
        u[0] = Fc;
#pragma unroll
        for (int k = 0; k < 3; ++k) u[k + 1] = (t1[k] + cr[k]) - t3[k];
        mat4(c.Binv, u, Td);
    } else if constexpr (ACT == VF_ACT_VELOCITY || ACT == VF_ACT_POSITION) {
        geometric_controller<ACT == VF_ACT_POSITION>(c, s, a, Td, vstrided);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) Td[k] = c.m * (a[k] * c.acc_half + c.acc_mean);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) Td[k] = clampf(Td[k], c.T_min, c.T_max);
}

// ---- the three recurrences of one sub-step (dynamics.py:335-367) ----------------------------------
// They only couple one way: motors -> (F, tau); rotation (q, w) needs tau; translation (p, v) needs
// q and F.  The fused kernels run them either in one thread or on two co-resident waves.

// rotor set-point contribution (1 - c) * omega_des: loop invariant (thrust_des is fixed for the
// interval), so the quadratic root of dynamics.py:545-553 is taken once, not once per sub-step.
template <bool CTRL_DELAY>
__device__ __forceinline__ void rotor_setpoint(const vf_dyn_cfg& c, const float* Td, float* wd)
{
    if constexpr (CTRL_DELAY) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d3 = c.rot_tm1sq - c.rot_4tm0 * (c.tm2 - Td[k]);
            wd[k] = (c.one_minus_c) * (c.rot_scale * (c.rot_neg_tm1 + vf_sqrt(d3)));
        }
    }
}

// _run_motors + allocation (:338-339,505-534): wm, T updated; ft = [F, tau]
template <bool CTRL_DELAY>
__device__ __forceinline__ void motor_substep(const vf_dyn_cfg& c, const float* Td, const float* wd, float* wm, float* T,
                                              float* ft)
{
    if constexpr (CTRL_DELAY) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wm[k] = c.c_motor * wm[k] + wd[k];                        // :514
            const float wp = wm[k] + 0.0f;
            T[k] = (c.tm0 * (wp * wp) + c.tm1 * wm[k]) + c.tm2;       // :530-534
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) T[k] = Td[k];                     // :518
    }
    mat4(c.B, T, ft);                                                  // :339
}

// linear acceleration from the state at the START of the sub-step (:342-347)
__device__ __forceinline__ void linear_acc(const vf_dyn_cfg& c, const Quat& q, const float* v, float F, const float* kl,
                                           const float* kq, float* acc)
{
    const Quat vq{0.0f, v[0] + 0.0f, v[1] + 0.0f, v[2] + 0.0f};
    const Quat vb = qmul(qmul(qconj(q), vq), q);
    const float vbv[3] = {vb.x, vb.y, vb.z};
    float u[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float drag = kl[k] * vbv[k] + (kq[k] * vbv[k]) * __builtin_fabsf(vbv[k]);
        const float zf = (k == 2 ? 1.0f : 0.0f) * F;
        u[k] = zf - drag;
    }
    const Quat uq{0.0f, u[0], u[1], u[2]};
    const Quat ra = qmul(qmul(q, uq), qconj(q));
#if VF_FAST_EXACT
    {   // wave-uniform guard: every numerator is 0 or has 2^-100 <= |x| <= 2^100; the sign of a zero quotient does not
        // survive the "+ 0.0f" / "+ g_z" below
        const int e0 = __builtin_amdgcn_frexp_expf(ra.x), e1 = __builtin_amdgcn_frexp_expf(ra.y), e2 = __builtin_amdgcn_frexp_expf(ra.z);
        const int emin = min(min(e0, e1), e2);
        const float amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(ra.x), __builtin_fabsf(ra.y)), __builtin_fabsf(ra.z));
        if (VF_FAST_EXACT == 2 || __builtin_amdgcn_ballot_w64(!(emin >= -99 && amax <= 0x1p100f)) == 0) {
            const float y = 1.0f / c.m;
            acc[0] = div_const_core(ra.x, c.m, y) + 0.0f;
            acc[1] = div_const_core(ra.y, c.m, y) + 0.0f;
            acc[2] = div_const_core(ra.z, c.m, y) + c.g_z;
            return;
        }
    }
#endif
    acc[0] = ra.x / c.m + 0.0f;
    acc[1] = ra.y / c.m + 0.0f;
    acc[2] = ra.z / c.m + c.g_z;
}

// translation: acc, then p and v advance (maths.py:310,344,346 / repaired rk4 :353-386)
template <int INTEG>
__device__ __forceinline__ void trans_substep(const vf_dyn_cfg& c, const Quat& q, float F, const float* kl, const float* kq,
                                              const float* wind, float* p, float* v, float* acc)
{
    linear_acc(c, q, v, F, kl, kq, acc);
    const float dt = c.dt;
    if constexpr (INTEG == VF_INT_EULER) {
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = p[k] + (v[k] + wind[k]) * dt;
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = v[k] + acc[k] * dt;
    } else {
        const float ks0 = 1.0f / 6.0f, ks1 = 2.0f / 6.0f;
        float sp[3], sv[3];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const float h = st == 3 ? 1.0f : 0.5f;
            const float ks = (st == 0 || st == 3) ? ks0 : ks1;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float vc = st == 0 ? v[k] : v[k] + acc[k] * h * dt;
                const float kp = (vc + wind[k]) * ks, kv = acc[k] * ks;
                sp[k] = st == 0 ? kp : sp[k] + kp;
                sv[k] = st == 0 ? kv : sv[k] + kv;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = p[k] + sp[k] * dt;
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = v[k] + sv[k] * dt;
    }
}

// rotation: q and w advance, aa = angular acceleration handed to the controller, q re-normalised
// (maths.py:311,314,345,347,351; dynamics.py:367; repaired rk4: SURVEY App. C-1 -- stages see the
// caller's wind, ks-contractions are ((k1*w0 + k2*w1) + k3*w2) + k4*w3, tau stays frozen)
template <int INTEG>
__device__ __forceinline__ void rot_substep(const vf_dyn_cfg& c, const float* tq, Quat& q, float* w, float* aa)
{
    const float dt = c.dt;
    if constexpr (INTEG == VF_INT_EULER) {
        float dq[4], dw[3];
        derivs(c, q, w, tq, dq, dw);
        q.w = q.w + dq[0] * dt;
        q.x = q.x + dq[1] * dt;
        q.y = q.y + dq[2] * dt;
        q.z = q.z + dq[3] * dt;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            w[k] = w[k] + dw[k] * dt;
            aa[k] = dw[k];
        }
    } else {
        const float ks0 = 1.0f / 6.0f, ks1 = 2.0f / 6.0f;
        Quat qc = q;
        float wc[3] = {w[0], w[1], w[2]};
        float sq[4], sw[3], dq[4], dw[3];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            if (st != 0) {
                const float h = st == 3 ? 1.0f : 0.5f;
                qc.w = q.w + dq[0] * h * dt;
                qc.x = q.x + dq[1] * h * dt;
                qc.y = q.y + dq[2] * h * dt;
                qc.z = q.z + dq[3] * h * dt;
#pragma unroll
                for (int k = 0; k < 3; ++k) wc[k] = w[k] + dw[k] * h * dt;
            }
            derivs(c, qc, wc, tq, dq, dw);
            const float ks = (st == 0 || st == 3) ? ks0 : ks1;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float kw = dw[k] * ks;
                sw[k] = st == 0 ? kw : sw[k] + kw;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float kq4 = dq[k] * ks;
                sq[k] = st == 0 ? kq4 : sq[k] + kq4;
            }
        }
        q.w = q.w + sq[0] * dt;
        q.x = q.x + sq[1] * dt;
        q.y = q.y + sq[2] * dt;
        q.z = q.z + sq[3] * dt;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            w[k] = w[k] + sw[k] * dt;
            aa[k] = sw[k];
        }
    }
    const float nn = vf_sqrt(((q.w * q.w + q.x * q.x) + q.y * q.y) + q.z * q.z);   // :367, maths.py:226-230
    q.w = q.w / nn;
    q.x = q.x / nn;
    q.y = q.y / nn;
    q.z = q.z / nn;
}

// t += ctrl_dt and the state clamps (_ugly_fix, dynamics.py:368-382)
__device__ __forceinline__ void finish_interval(const vf_dyn_cfg& c, Agent& s)
{
    s.t = s.t + c.ctrl_dt;
    s.p[0] = clampf(s.p[0], -c.pos_xy_lim, c.pos_xy_lim);
    s.p[1] = clampf(s.p[1], -c.pos_xy_lim, c.pos_xy_lim);
    s.p[2] = clampf(s.p[2], c.pos_z_lo, c.pos_z_hi);
#pragma unroll
    for (int k = 0; k < 3; ++k) s.v[k] = clampf(s.v[k], -c.vel_lim, c.vel_lim);
#pragma unroll
    for (int k = 0; k < 3; ++k) s.w[k] = clampf(s.w[k], -c.omg_lim, c.omg_lim);
}

// Optional observer of a control interval: head(sub, s) sees the agent at the head of every sub-step, end(s) the state after the last
// one BEFORE finish_interval clamps it.  k_bptt_rollout (vf_bptt_rollout.hip) writes them to the sub-step tape that lets the
// persistent reverse sweep skip its replay of the interval (vf_env_bwd_body.hpp).
struct NoCheckpoint {
    __device__ __forceinline__ void head(int, const Agent&) const {}
    __device__ __forceinline__ void end(const Agent&) const {}
};

// All sub-steps of one control interval in ONE thread (dynamics.py:335-382).  kl/kq: this agent's drag.
template <int ACT, int INTEG, bool CTRL_DELAY, class CK = NoCheckpoint>
__device__ __forceinline__ void control_interval(const vf_dyn_cfg& c, Agent& s, const float* a,
                                                 const float* kl, const float* kq, bool vstrided = false, const CK& ck = CK{})
{
    float Td[4], wd[4];
    desired_thrusts<ACT>(c, s, a, Td, vstrided);
    rotor_setpoint<CTRL_DELAY>(c, Td, wd);
#if VF_SUBSTEP_UNROLL == 2
#pragma unroll 2
#elif VF_SUBSTEP_UNROLL == 4
#pragma unroll 4
#elif VF_SUBSTEP_UNROLL == 8
#pragma unroll 8
#else
#pragma unroll 1
#endif
    for (int sub = 0; sub < c.interval_steps; ++sub) {
        ck.head(sub, s);
        float ft[4];
        motor_substep<CTRL_DELAY>(c, Td, wd, s.wm, s.T, ft);
        trans_substep<INTEG>(c, s.q, ft[0], kl, kq, s.wnd, s.p, s.v, s.acc);   // uses q of the sub-step start
        rot_substep<INTEG>(c, ft + 1, s.q, s.w, s.aa);
    }
    ck.end(s);
    finish_interval(c, s);
}

// ---- global stores of the step kernels ----
// A step leaves ~200 B per agent dirty (13 MB at 65 536 agents).  With plain stores those lines sit in the XCD L2s until the
// end-of-kernel release writes them back, which the NEXT launch waits for (MI355X_MICROARCH.md "boundary": + B / 6 TB/s when
// the predecessor leaves B bytes dirty); write-through stores (sc1) push them out while the other waves still compute.
// VF_STORE_MODE: 0 plain, 1 sc1 (write-through), 2 nt, 3 sc0 sc1 -- A/B knob of tools/env_step_probe.hip.
#ifndef VF_STORE_MODE
#define VF_STORE_MODE 0
#endif
typedef float vf_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4(float4* p, const float4 v)
{
#if VF_STORE_MODE != 0
    const vf_f4 x = {v.x, v.y, v.z, v.w};
#endif
#if VF_STORE_MODE == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(x) : "memory");
#elif VF_STORE_MODE == 2
    asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(x) : "memory");
#elif VF_STORE_MODE == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(x) : "memory");
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st1(float* p, const float v)
{
#if VF_STORE_MODE == 1
    asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
#elif VF_STORE_MODE == 2
    asm volatile("global_store_dword %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
#elif VF_STORE_MODE == 3
    asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
#else
    *p = v;
#endif
}

// store with an explicit cache policy (store_rows_coalesced's MODE; measured in profiles/r04_env_quad.txt): 0 plain, 1 sc1 (write-through), 2 nt, 3 sc0 sc1
template <int MODE>
__device__ __forceinline__ void st4_mode(float4* p, const float4 v)
{
    if constexpr (MODE == 0) {
        st4(p, v);
    } else {
        const vf_f4 x = {v.x, v.y, v.z, v.w};
        if constexpr (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(x) : "memory");
        else if constexpr (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(x) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(x) : "memory");
    }
}
template <int MODE>
__device__ __forceinline__ void st1_mode(float* p, const float v)
{
    if constexpr (MODE == 0) st1(p, v);
    else if constexpr (MODE == 1) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    else if constexpr (MODE == 2) asm volatile("global_store_dword %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}

// ---- slab I/O: wave-tile AoSoA, one 16-byte granule per lane per access ----
// float4 address of (agent i, granule g): see include/visfly_amd.h
__device__ __forceinline__ float4* granule(float* __restrict__ S, int G, int i, int g)
{
    return reinterpret_cast<float4*>(S) + ((size_t)(i >> 6) * G + g) * VF_TILE + (i & 63);
}

struct Spares {  // component 0 of the vector granules (env-layer slots), carried through untouched
    float vel, omg, aacc, acc;
};

// LOAD_THRUSTS = false: the step kernels never read the rotor thrusts of the previous interval (every sub-step
// recomputes them from the rotor speeds, or takes the set-point when ctrl_delay is off: dynamics.py:505-534), so that
// granule is written but not fetched -- 16 of the 160 B an agent-step used to load.
template <bool LOAD_THRUSTS = true>
__device__ __forceinline__ void load_agent(float* __restrict__ S, int G, int i, Agent& s, Spares& sp)
{
    // issue order = order of first use: ring head (velocity granule) and the controller's inputs (body rates, angular
    // acceleration) first, then the rotors, then what the translation needs -- loads return in order, so the controller
    // and the ring exchange run while the tail of the burst is still in flight
    s.wnd[0] = s.wnd[1] = s.wnd[2] = 0.0f;      // set by load_wind where the wind matters
    const float4 g2 = *granule(S, G, i, VF_G_VEL);
    const float4 g3 = *granule(S, G, i, VF_G_OMG);
    const float4 g6 = *granule(S, G, i, VF_G_AACC);
    const float4 g4 = *granule(S, G, i, VF_G_MOT);
    const float4 g1 = *granule(S, G, i, VF_G_QUAT);
    const float4 g0 = *granule(S, G, i, VF_G_POS);
    float4 g5 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (LOAD_THRUSTS) g5 = *granule(S, G, i, VF_G_THR);
    const float4 g7 = *granule(S, G, i, VF_G_ACC);
    s.t = g0.x; s.p[0] = g0.y; s.p[1] = g0.z; s.p[2] = g0.w;
    s.q = Quat{g1.x, g1.y, g1.z, g1.w};
    sp.vel = g2.x; s.v[0] = g2.y; s.v[1] = g2.z; s.v[2] = g2.w;
    sp.omg = g3.x; s.w[0] = g3.y; s.w[1] = g3.z; s.w[2] = g3.w;
    s.wm[0] = g4.x; s.wm[1] = g4.y; s.wm[2] = g4.z; s.wm[3] = g4.w;
    s.T[0] = g5.x; s.T[1] = g5.y; s.T[2] = g5.z; s.T[3] = g5.w;
    sp.aacc = g6.x; s.aa[0] = g6.y; s.aa[1] = g6.z; s.aa[2] = g6.w;
    sp.acc = g7.x; s.acc[0] = g7.y; s.acc[1] = g7.z; s.acc[2] = g7.w;
}

__device__ __forceinline__ void store_agent(float* __restrict__ S, int G, int i, const Agent& s, const Spares& sp)
{
    st4(granule(S, G, i, VF_G_POS), make_float4(s.t, s.p[0], s.p[1], s.p[2]));
    st4(granule(S, G, i, VF_G_QUAT), make_float4(s.q.w, s.q.x, s.q.y, s.q.z));
    st4(granule(S, G, i, VF_G_VEL), make_float4(sp.vel, s.v[0], s.v[1], s.v[2]));
    st4(granule(S, G, i, VF_G_OMG), make_float4(sp.omg, s.w[0], s.w[1], s.w[2]));
    st4(granule(S, G, i, VF_G_MOT), make_float4(s.wm[0], s.wm[1], s.wm[2], s.wm[3]));
    st4(granule(S, G, i, VF_G_THR), make_float4(s.T[0], s.T[1], s.T[2], s.T[3]));
    st4(granule(S, G, i, VF_G_AACC), make_float4(sp.aacc, s.aa[0], s.aa[1], s.aa[2]));
    st4(granule(S, G, i, VF_G_ACC), make_float4(sp.acc, s.acc[0], s.acc[1], s.acc[2]));
}

struct DynArgs {
    int N;      // live agents
    int G;      // granules per agent
    int g_drag; // first drag granule or -1
    float* S;   // slab
    const float4* action;  // (N,4)
    float* obs;            // (N,13) or null
    int head;              // delay-ring slot of this launch (= every agent's head word; vf_handles.hpp)
    const float4* wind = nullptr;   // per-agent wind of this control interval (N rows x,y,z,-; vf_*_set_wind) or null = vf_dyn_cfg.wind
    int vstrided = 0;               // 1: the reference's velocity tensor is a strided view (vf_dyn::vel_strided; geometric_controller)
};

// wind velocity of the interval (dynamics.py:320,384-388: update_wind() runs first in step(), the value holds for all sub-steps)
__device__ __forceinline__ void load_wind(const vf_dyn_cfg& c, const DynArgs& g, int i, bool live, Agent& s)
{
    s.wnd[0] = c.wind[0]; s.wnd[1] = c.wind[1]; s.wnd[2] = c.wind[2];
    if (g.wind && live) {
        const float4 w = g.wind[i];
        s.wnd[0] = w.x; s.wnd[1] = w.y; s.wnd[2] = w.z;
    }
}

// Pops the oldest action of agent i from its ring slot and pushes the new one (dynamics.py:323-328).  The slot index is
// launch-uniform (g.head = control steps since the last full reset mod delay_steps, kept by the host handle): the address is
// known before any state arrives, so the slot load travels with the first burst instead of waiting for the velocity
// granule.  The per-agent head word in that granule's spare component is still advanced -- the adjoint kernel reads it
// from its tape -- and a reset zeroes all slots, after which any head position is equivalent.
// an_reg: the new action in a register (a persistent launch that just computed it) instead of g.action[i]
// D: c.delay_steps -- a parameter of its own for k_env_step, which has it as a preloaded kernel argument (the slot load then does not
// wait for the constant block)
__device__ __forceinline__ void ring_exchange_d(const DynArgs& g, int i, bool live, float& head_bits, float* a, const float4* an_reg, int D)
{
    float4 an = make_float4(0.f, 0.f, 0.f, 0.f);
    if (an_reg) an = *an_reg;
    else if (live) an = g.action[i];
    if (D > 0) {
        const int head = g.head;
        float4* slot = granule(g.S, g.G, i, VF_G_RING + head);
        const float4 old = *slot;
        st4(slot, an);
        an = old;
        head_bits = __int_as_float(head + 1 == D ? 0 : head + 1);
    }
    a[0] = an.x; a[1] = an.y; a[2] = an.z; a[3] = an.w;
}
__device__ __forceinline__ void ring_exchange(const vf_dyn_cfg& c, const DynArgs& g, int i, bool live, float& head_bits,
                                              float* a, const float4* an_reg = nullptr)
{
    ring_exchange_d(g, i, live, head_bits, a, an_reg, c.delay_steps);
}

__device__ __forceinline__ void drag_of(const vf_dyn_cfg& c, const DynArgs& g, int i, float* kl, float* kq)
{
    if (g.g_drag >= 0) {
        const float4 a = *granule(g.S, g.G, i, g.g_drag), b = *granule(g.S, g.G, i, g.g_drag + 1);
        kl[0] = a.y; kl[1] = a.z; kl[2] = a.w;
        kq[0] = b.y; kq[1] = b.z; kq[2] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { kl[k] = c.k_lin[k]; kq[k] = c.k_quad[k]; }
    }
}

// state(N,13) row of one agent: [p, q wxyz, v + wind, w]  (dynamics.py:779-786)
__device__ __forceinline__ void obs_row(const vf_dyn_cfg& c, const Agent& s, float* o)
{
    o[0] = s.p[0]; o[1] = s.p[1]; o[2] = s.p[2];
    o[3] = s.q.w; o[4] = s.q.x; o[5] = s.q.y; o[6] = s.q.z;
    o[7] = s.v[0] + s.wnd[0]; o[8] = s.v[1] + s.wnd[1]; o[9] = s.v[2] + s.wnd[2];
    o[10] = s.w[0]; o[11] = s.w[1]; o[12] = s.w[2];
}

// AoS (N,C) output through LDS so that the global stores are coalesced dwords.  Wave-local: lane l parks
// its C-float row at tile[l*C..] (C odd -> conflict-free), then the wave streams its 64 rows out linearly.
// No workgroup barrier: one wave's LDS operations execute in order.  `tile` holds 64*C floats per wave.
template <int C, int MODE = 0>
__device__ __forceinline__ void store_rows_coalesced(float* __restrict__ out, int N, int wave_first, const float* row,
                                                     float* tile)
{
    const int l = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < C; ++k) tile[l * C + k] = row[k];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    const int rows_here = min(64, N - wave_first);
    float* dst = out + (size_t)wave_first * C;
    // a full wave's 64 rows are 64 * C contiguous floats starting at a 16-byte boundary (64 * C * 4 B per wave): stream them
    // out as 16-byte stores -- C dwordx4 per 4 lanes instead of C dword stores per lane (the tail of the launch is bound by
    // the number of store instructions, not by their bytes)
    if (rows_here == 64 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        constexpr int Q = 16 * C;   // float4 per wave
#pragma unroll
        for (int k = 0; k < (Q + 63) / 64; ++k) {
            const int j = k * 64 + l;
            if (j < Q) st4_mode<MODE>(reinterpret_cast<float4*>(dst) + j, *reinterpret_cast<const float4*>(tile + 4 * j));
        }
        return;
    }
    const int total = rows_here * C;
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const int j = k * 64 + l;
        if (j < total) st1_mode<MODE>(dst + j, tile[j]);
    }
}

// ... for a wave whose QUADS of lanes hold one agent each (16 rows per wave): lane 4 m + k parks row m (the four lanes of a quad hold the
// same row: same values to the same addresses), then the wave streams its <= 16 rows out
template <int C>
__device__ __forceinline__ void store_rows_quads(float* __restrict__ out, int N, int wave_first, const float* row, float* tile)
{
    const int l = threadIdx.x & 63, m = l >> 2;
#pragma unroll
    for (int k = 0; k < C; ++k) tile[m * C + k] = row[k];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    const int total = min(16, N - wave_first) * C;
    float* dst = out + (size_t)wave_first * C;
#pragma unroll
    for (int k = 0; k < (16 * C + 63) / 64; ++k) {
        const int j = k * 64 + l;
        if (j < total) st1(dst + j, tile[j]);
    }
}

// ---- two-wave split of the control interval --------------------------------------------------------
// One wave per SIMD cannot hide its own dependent-issue latency (measured 4.6 cycles / instruction,
// 2.5 with four waves per SIMD).  Rotation (motors -> torque -> q, w) never reads the translational
// state, so a 128-thread workgroup runs the two recurrences of the SAME 64 agents on two waves, the
// translation wave one hand-off behind: per sub-step the rotation wave publishes (q, F) through LDS.
struct __attribute__((aligned(16))) SplitShared {
    float xq[2][5][64];   // double-buffered (q.w, q.x, q.y, q.z, F) of the sub-step start
    float fin[20][64];    // final q, w, wm, T, aa + the two env spares the rotation wave loaded
    float tile[64 * 13];  // observation transpose of the translation wave
};

__device__ __forceinline__ void lds_publish_barrier()
{
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
}

// rotation wave: everything that does not need p / v.  Returns after the final hand-off.
// D: c.delay_steps when the caller has it as a preloaded kernel argument (k_env_step_split), else -1
// PFK > 0: behind the first loads, one batch of scalar loads touches the PFK bytes of kernel arguments and the lines of the constant
// blocks pf_a (vf_dyn_cfg) / pf_b (first two lines) + the word at pf_c (prefetch_kernarg_and_const_lines, vf_common.hpp)
template <int ACT, int INTEG, bool CTRL_DELAY, int PFK = 0>
__device__ __forceinline__ void split_rotation_wave(const vf_dyn_cfg& c, const DynArgs& g, int i, bool live, SplitShared& sh, int D = -1,
                                                    const void* pf_b = nullptr, const void* pf_c = nullptr)
{
    const int l = threadIdx.x & 63;
    const float4 g1 = *granule(g.S, g.G, i, VF_G_QUAT), g3 = *granule(g.S, g.G, i, VF_G_OMG);
    const float4 g4 = *granule(g.S, g.G, i, VF_G_MOT), g5 = *granule(g.S, g.G, i, VF_G_THR);
    const float4 g6 = *granule(g.S, g.G, i, VF_G_AACC);
    float head_bits = 0.0f;   // out-parameter only: the translation wave owns the head word
    Agent s;
    s.q = Quat{g1.x, g1.y, g1.z, g1.w};
    s.w[0] = g3.y; s.w[1] = g3.z; s.w[2] = g3.w;
    s.wm[0] = g4.x; s.wm[1] = g4.y; s.wm[2] = g4.z; s.wm[3] = g4.w;
    s.T[0] = g5.x; s.T[1] = g5.y; s.T[2] = g5.z; s.T[3] = g5.w;
    s.aa[0] = g6.y; s.aa[1] = g6.z; s.aa[2] = g6.w;
    float a[4];
    if (D >= 0) ring_exchange_d(g, i, live, head_bits, a, nullptr, D);   // pushes the new action; the translation wave advances the head word
    else ring_exchange(c, g, i, live, head_bits, a);
    if constexpr (PFK > 0) prefetch_kernarg_and_const_lines<PFK, (sizeof(vf_dyn_cfg) + 63) / 64, 2>(&c, pf_b, pf_c);
    float Td[4], wd[4];
    desired_thrusts<ACT>(c, s, a, Td);
    rotor_setpoint<CTRL_DELAY>(c, Td, wd);
    for (int sub = 0; sub < c.interval_steps; ++sub) {
        float ft[4];
        motor_substep<CTRL_DELAY>(c, Td, wd, s.wm, s.T, ft);
        float(*x)[64] = sh.xq[sub & 1];
        x[0][l] = s.q.w; x[1][l] = s.q.x; x[2][l] = s.q.y; x[3][l] = s.q.z; x[4][l] = ft[0];
        lds_publish_barrier();
        rot_substep<INTEG>(c, ft + 1, s.q, s.w, s.aa);
    }
    float(*f)[64] = sh.fin;
    f[0][l] = s.q.w; f[1][l] = s.q.x; f[2][l] = s.q.y; f[3][l] = s.q.z;
#pragma unroll
    for (int k = 0; k < 3; ++k) { f[4 + k][l] = s.w[k]; f[15 + k][l] = s.aa[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { f[7 + k][l] = s.wm[k]; f[11 + k][l] = s.T[k]; }
    f[18][l] = g3.x;   // spare of the omega granule (env: step counter)
    f[19][l] = g6.x;   // spare of the angular-acceleration granule (env: reward sum)
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): the ring push is globally performed before the
    __builtin_amdgcn_s_barrier();        // translation wave may overwrite the same slot (reset) after this barrier
}

// translation wave: p, v, acc through the sub-steps, then assembles the full agent state (clamped, t advanced)
template <int INTEG, int PFK = 0>
__device__ __forceinline__ void split_translation_wave(const vf_dyn_cfg& c, const DynArgs& g, int i, SplitShared& sh, Agent& s,
                                                       Spares& sp, int D = -1, const void* pf_b = nullptr, const void* pf_c = nullptr)
{
    const int l = threadIdx.x & 63;
    const float4 g0 = *granule(g.S, g.G, i, VF_G_POS), g2 = *granule(g.S, g.G, i, VF_G_VEL);
    const float4 g7 = *granule(g.S, g.G, i, VF_G_ACC);
    if constexpr (PFK > 0) prefetch_kernarg_and_const_lines<PFK, (sizeof(vf_dyn_cfg) + 63) / 64, 2>(&c, pf_b, pf_c);
    float kl[3], kq[3];
    drag_of(c, g, i, kl, kq);
    load_wind(c, g, i, i < g.N, s);
    s.t = g0.x; s.p[0] = g0.y; s.p[1] = g0.z; s.p[2] = g0.w;
    s.v[0] = g2.y; s.v[1] = g2.z; s.v[2] = g2.w;
    s.acc[0] = g7.y; s.acc[1] = g7.z; s.acc[2] = g7.w;
    sp.acc = g7.x;
    sp.vel = g2.x;
    const int Dd = D >= 0 ? D : c.delay_steps;
    if (Dd > 0)    // same head update ring_exchange applies (the rotation wave did the exchange itself)
        sp.vel = __int_as_float(g.head + 1 == Dd ? 0 : g.head + 1);
    for (int sub = 0; sub < c.interval_steps; ++sub) {
        __builtin_amdgcn_s_barrier();
        const float(*x)[64] = sh.xq[sub & 1];
        const Quat q{x[0][l], x[1][l], x[2][l], x[3][l]};
        const float F = x[4][l];
        trans_substep<INTEG>(c, q, F, kl, kq, s.wnd, s.p, s.v, s.acc);
    }
    __builtin_amdgcn_s_barrier();
    const float(*f)[64] = sh.fin;
    s.q = Quat{f[0][l], f[1][l], f[2][l], f[3][l]};
#pragma unroll
    for (int k = 0; k < 3; ++k) { s.w[k] = f[4 + k][l]; s.aa[k] = f[15 + k][l]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s.wm[k] = f[7 + k][l]; s.T[k] = f[11 + k][l]; }
    sp.omg = f[18][l];
    sp.aacc = f[19][l];
    finish_interval(c, s);
}

}  // namespace vf
