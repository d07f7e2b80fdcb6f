/*
 * vf_oracle.c -- TEST INFRASTRUCTURE ONLY (see vf_oracle.h header).
 *
 * Plain-C fp32 restatement of the VisFly hot path.  Build with
 *   gcc -O2 -ffp-contract=off -fno-fast-math -mfma -fopenmp
 * Every elementwise torch op is one separately rounded fp32 operation; the
 * only fused operations are the ones torch itself fuses on the oracle host
 * (SURVEY App. B.4): BLAS matmuls with K in {3,4} are k-ordered FMA chains and
 * torch.linalg.cross is fma(a_i, b_j, -(a_j*b_i)).  Those sites call fmaf()
 * explicitly; everything else relies on -ffp-contract=off.
 */
#include "vf_oracle.h"
#include "vf_sleef.h"   /* atan2 = SLEEF u10 restated; all SLEEF u10 restated (see its header) */

#include <math.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* contiguous agent range of the calling OpenMP thread (whole range outside a parallel region) */
static inline void thread_range(int N, int* i0, int* i1)
{
#ifdef _OPENMP
    const int nt = omp_get_num_threads(), t = omp_get_thread_num();
#else
    const int nt = 1, t = 0;
#endif
    const int per = (N + nt - 1) / nt;
    *i0 = t * per < N ? t * per : N;
    *i1 = *i0 + per < N ? *i0 + per : N;
}

void vfo_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int vfo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------- small helpers ---------- */

/* (3x3) @ v as the k-ordered FMA chain torch's sgemm produces (App. B.4) */
static inline void mat3_chain(const float* A, float x0, float x1, float x2, float* o)
{
    for (int i = 0; i < 3; ++i) {
        float acc = A[3 * i + 0] * x0;
        acc = fmaf(A[3 * i + 1], x1, acc);
        acc = fmaf(A[3 * i + 2], x2, acc);
        o[i] = acc;
    }
}

/* (4x4) @ v, same chain */
static inline void mat4_chain(const float* A, const float* x, float* o)
{
    for (int i = 0; i < 4; ++i) {
        float acc = A[4 * i + 0] * x[0];
        acc = fmaf(A[4 * i + 1], x[1], acc);
        acc = fmaf(A[4 * i + 2], x[2], acc);
        acc = fmaf(A[4 * i + 3], x[3], acc);
        o[i] = acc;
    }
}

/* Hamilton product, term order and rounding of utils/maths.py:168-174 */
typedef struct { float w, x, y, z; } quat;

static inline quat qmul(quat a, quat b)
{
    quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return r;
}

static inline quat qconj(quat a) /* utils/maths.py:232-233 */
{
    quat r = { a.w, -a.x, -a.y, -a.z };
    return r;
}

static inline float clampf(float v, float lo, float hi) /* th.clamp semantics */
{
    /* torch clamp: min(max(v, lo), hi); NaN propagates */
    float r = v < lo ? lo : v;
    r = r > hi ? hi : r;
    return r;
}

/* derivatives of (q, omega) used by both integrators (utils/maths.py:300-315) */
static inline void derivs(const vfo_consts* c, quat q, const float* w, const float* tq,
                          float* dq, float* dw)
{
    /* d_q = (ori * Quaternion(0, *ori_vel) * 0.5)          maths.py:311 */
    quat wq = { 0.0f, w[0], w[1], w[2] };
    quat p = qmul(q, wq);
    dq[0] = p.w * 0.5f; dq[1] = p.x * 0.5f; dq[2] = p.y * 0.5f; dq[3] = p.z * 0.5f;
    /* d_ori_vel = J_inv @ (tau - linalg.cross(w, J @ w))    maths.py:314 */
    float Jw[3];
    mat3_chain(c->J, w[0], w[1], w[2], Jw);
    float cr[3];
    cr[0] = fmaf(w[1], Jw[2], -(w[2] * Jw[1]));
    cr[1] = fmaf(w[2], Jw[0], -(w[0] * Jw[2]));
    cr[2] = fmaf(w[0], Jw[1], -(w[1] * Jw[0]));
    float r0 = tq[0] - cr[0], r1 = tq[1] - cr[1], r2 = tq[2] - cr[2];
    mat3_chain(c->Jinv, r0, r1, r2, dw);
}


/* Geometric SO(3) controller of the velocity / position action types
 * (envs/base/dynamics.py:414-452 velocity, :453-496 position), once per control
 * interval.  x.norm(dim=0) over (2,N)/(3,N) rows = FMA chain + IEEE sqrt [probe];
 * cross() is the helper of utils/maths.py:392-394 (separately rounded, "+ 0");
 * 3x3 @ 3x3 and 3x3 @ (3,N) are k-ordered FMA chains (App. B.4).  sin/cos/atan2
 * are the libm ones: torch's vectorised SLEEF variants are not reproducible, so
 * parity for these two action types is tolerance-level, not bit-level. */
static inline void cross_helper(const float* a, const float* b, float* o)
{
    o[0] = (a[1] * b[2] - a[2] * b[1]) + 0.0f;
    o[1] = (a[2] * b[0] - a[0] * b[2]) + 0.0f;
    o[2] = (a[0] * b[1] - a[1] * b[0]) + 0.0f;
}

static void geometric_controller(const vfo_consts* c, const float* a, const float* p, quat q,
                                 const float* v, const float* w, const float* al, float* Td, int strided_v)
{
    const int pos_mode = c->action_type == VFO_ACT_POSITION;
    /* _de_normalize :716-730 -> [yaw, x, y, z] */
    float cmd[4];
    cmd[0] = a[0] * c->yaw_half + c->yaw_mean;
    for (int k = 1; k < 4; ++k) cmd[k] = a[k] * c->vel_half + c->vel_mean;
    float F[3];
    for (int k = 0; k < 3; ++k) {
        float a_des;
        if (pos_mode) {
            float v_des = c->pos_d * (cmd[k + 1] - p[k]);      /* :456 */
            a_des = c->vel_d * (v_des - v[k]);                 /* :457 */
        } else {
            a_des = c->vel_p * (cmd[k + 1] - v[k]);            /* :416 */
        }
        F[k] = c->m * (a_des - (k == 2 ? c->g_z : 0.0f));      /* :417,458 */
    }
    /* Quaternion.toEuler()[2] (utils/maths.py:248) */
    float yaw_cur = vfs_atan2f_u10(2.0f * (q.w * q.z + q.x * q.y), 1.0f - 2.0f * (q.y * q.y + q.z * q.z));
    float yaw_des, gain;
    if (pos_mode) {
        yaw_des = cmd[0];                                      /* :461 */
        gain = c->pos_d;                                       /* :468 */
    } else {
        float vn = sqrtf(fmaf(v[1], v[1], v[0] * v[0]));       /* :421 */
        /* :423-427; torch.atan2 on strided velocity rows = glibc's atan2f, on contiguous ones SLEEF's (vf_sleef.h) */
        yaw_des = vn > 0.1f ? (strided_v ? vfs_atan2f_glibc(v[1], v[0]) : vfs_atan2f_u10(v[1], v[0])) : yaw_cur;
        gain = c->vel_d;                                       /* :433 */
    }
    float ye = yaw_des - yaw_cur;
    const int crm = c->trig_mode == 1;
    ye = crm ? vfs_atan2f_u10(vfs_sinf_cr(ye), vfs_cosf_cr(ye)) : vfs_atan2f_u10(vfs_sinf_u10(ye), vfs_cosf_u10(ye));   /* :432,467 */
    float yaw_spd = ye * gain * 2.0f;
    /* gross thrust = (conj(q) * (0,F) * q).imag[2]            :435, maths.py:49,103 */
    quat fq = { 0.0f, F[0], F[1], F[2] };
    quat fb = qmul(qmul(qconj(q), fq), q);
    float gross = fb.z;
    /* Quaternion.R (utils/maths.py:116-120) */
    float R[3][3];
    R[0][0] = 1.0f - 2.0f * (q.y * q.y + q.z * q.z); R[0][1] = 2.0f * (q.x * q.y - q.z * q.w); R[0][2] = 2.0f * (q.x * q.z + q.y * q.w);
    R[1][0] = 2.0f * (q.x * q.y + q.z * q.w); R[1][1] = 1.0f - 2.0f * (q.x * q.x + q.z * q.z); R[1][2] = 2.0f * (q.y * q.z - q.x * q.w);
    R[2][0] = 2.0f * (q.x * q.z - q.y * q.w); R[2][1] = 2.0f * (q.y * q.z + q.x * q.w); R[2][2] = 1.0f - 2.0f * (q.x * q.x + q.y * q.y);
    /* desired frame :437-442 */
    float fn = sqrtf(fmaf(F[2], F[2], fmaf(F[1], F[1], F[0] * F[0])));
    float b3[3] = { F[0] / fn, F[1] / fn, F[2] / fn };
    float c1[3] = { crm ? vfs_cosf_cr(yaw_des) : vfs_cosf_u10(yaw_des), crm ? vfs_sinf_cr(yaw_des) : vfs_sinf_u10(yaw_des), 0.0f };
    float b2[3], b1[3];
    cross_helper(b3, c1, b2);
    float bn = sqrtf(fmaf(b2[2], b2[2], fmaf(b2[1], b2[1], b2[0] * b2[0])));
    for (int k = 0; k < 3; ++k) b2[k] = b2[k] / bn;
    cross_helper(b2, b3, b1);
    float Rd[3][3];                                            /* columns b1, b2, b3 */
    for (int r = 0; r < 3; ++r) { Rd[r][0] = b1[r]; Rd[r][1] = b2[r]; Rd[r][2] = b3[r]; }
    /* per-agent loop body :446-450: A = Rd^T R, Bm = R^T Rd, m = 0.5 (A - Bm) */
    float A[3][3], Bm[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = Rd[0][i] * R[0][j]; s = fmaf(Rd[1][i], R[1][j], s); s = fmaf(Rd[2][i], R[2][j], s);
            A[i][j] = s;
            float u = R[0][i] * Rd[0][j]; u = fmaf(R[1][i], Rd[1][j], u); u = fmaf(R[2][i], Rd[2][j], u);
            Bm[i][j] = u;
        }
    float m12 = 0.5f * (A[1][2] - Bm[1][2]), m02 = 0.5f * (A[0][2] - Bm[0][2]), m01 = 0.5f * (A[0][1] - Bm[0][1]);
    float pose[3] = { -(-m12), -m02, -(-m01) };
    float ang[3];
    for (int i = 0; i < 3; ++i) {
        float s = A[i][0] * 0.0f; s = fmaf(A[i][1], 0.0f, s); s = fmaf(A[i][2], yaw_spd, s);
        ang[i] = s - w[i];
    }
    float t1[3], t2[3], inner[3], tau[3], cr[3];
    mat3_chain(c->Pm, pose[0], pose[1], pose[2], t1);
    if (pos_mode) {
        /* :489-494 */
        float t3[3], Jw[3];
        mat3_chain(c->P12, ang[0], ang[1], ang[2], t2);
        mat3_chain(c->Dm, al[0], al[1], al[2], t3);
        mat3_chain(c->J, w[0], w[1], w[2], Jw);
        cross_helper(w, Jw, cr);
        for (int k = 0; k < 3; ++k) inner[k] = ((t1[k] + t2[k]) - t3[k]) - cr[k];
    } else {
        /* :451 */
        mat3_chain(c->Pm, ang[0], ang[1], ang[2], t2);
        cross_helper(w, w, cr);
        for (int k = 0; k < 3; ++k) inner[k] = (t1[k] + t2[k]) - cr[k];
    }
    mat3_chain(c->J, inner[0], inner[1], inner[2], tau);
    float u[4] = { gross, tau[0], tau[1], tau[2] };
    mat4_chain(c->Binv, u, Td);                                /* :453,496 */
}

/* ---------- Dynamics.step ---------- */

/* bit 30 of *tick (the ring head itself is *tick & ~VFO_TICK_VSTRIDED): the reference's `_velocity` tensor is a strided view,
 * i.e. the last full reset was given velocities (dynamics.py:236 stores vel.T; in-place updates and clamp() keep the layout).
 * It decides which atan2 torch runs for the velocity action type's auto-yaw: glibc's (strided operands) or SLEEF's. */
#define VFO_TICK_VSTRIDED (1 << 30)

static void dyn_step_range(const vfo_consts* c, int N, float* S, float* Q, int slot_vstr,
                           const float* klin, const float* kquad,
                           const float* action, float* obs, int i0, int i1)
{
    const int D = c->delay_steps;
    const float dt = c->dt;
    const int slot = slot_vstr & ~VFO_TICK_VSTRIDED, vstr = (slot_vstr & VFO_TICK_VSTRIDED) != 0;

    for (int i = i0; i < i1; ++i) {
#define ROW(r) S[(size_t)(r) * N + i]
        /* ---- delay queue: use oldest, store newest   dynamics.py:323-328 ---- */
        float a[4];
        if (D > 0) {
            float* qs = Q + ((size_t)slot * 4) * N + i;
            for (int k = 0; k < 4; ++k) {
                a[k] = qs[(size_t)k * N];
                qs[(size_t)k * N] = action[4 * (size_t)i + k];
            }
        } else {
            for (int k = 0; k < 4; ++k) a[k] = action[4 * (size_t)i + k];
        }

        float p[3] = { ROW(VFO_POS), ROW(VFO_POS + 1), ROW(VFO_POS + 2) };
        quat q = { ROW(VFO_QUAT), ROW(VFO_QUAT + 1), ROW(VFO_QUAT + 2), ROW(VFO_QUAT + 3) };
        float v[3] = { ROW(VFO_VEL), ROW(VFO_VEL + 1), ROW(VFO_VEL + 2) };
        float w[3] = { ROW(VFO_OMG), ROW(VFO_OMG + 1), ROW(VFO_OMG + 2) };
        float wm[4] = { ROW(VFO_MOT), ROW(VFO_MOT + 1), ROW(VFO_MOT + 2), ROW(VFO_MOT + 3) };
        float T[4] = { ROW(VFO_THR), ROW(VFO_THR + 1), ROW(VFO_THR + 2), ROW(VFO_THR + 3) };
        float al[3] = { ROW(VFO_AACC), ROW(VFO_AACC + 1), ROW(VFO_AACC + 2) };
        float acc[3] = { ROW(VFO_ACC), ROW(VFO_ACC + 1), ROW(VFO_ACC + 2) };
        float kl[3], kq[3];
        for (int k = 0; k < 3; ++k) {
            kl[k] = klin ? klin[(size_t)k * N + i] : c->k_lin[k];
            kq[k] = kquad ? kquad[(size_t)k * N + i] : c->k_quad[k];
        }

        /* ---- de-normalise + low-level controller (once per interval) ---- */
        float Td[4];
        if (c->action_type == VFO_ACT_BODYRATE) {
            /* dynamics.py:704-710 */
            float Fc = (a[0] * c->acc_half + c->acc_mean) * c->m;
            float wc[3];
            for (int k = 0; k < 3; ++k) wc[k] = a[k + 1] * c->rate_half + c->rate_mean;
            /* dynamics.py:401-413 */
            float e[3] = { wc[0] - w[0], wc[1] - w[1], wc[2] - w[2] };
            float t1[3], Jw[3], t3[3];
            mat3_chain(c->JP, e[0], e[1], e[2], t1);
            float w0 = w[0] + 0.0f, w1 = w[1] + 0.0f, w2 = w[2] + 0.0f;
            mat3_chain(c->J, w0, w1, w2, Jw);
            /* cross(): utils/maths.py:392-394, separately rounded, "+ 0" */
            float cr[3];
            cr[0] = (w1 * Jw[2] - w2 * Jw[1]) + 0.0f;
            cr[1] = (w2 * Jw[0] - w0 * Jw[2]) + 0.0f;
            cr[2] = (w0 * Jw[1] - w1 * Jw[0]) + 0.0f;
            mat3_chain(c->Dm, al[0], al[1], al[2], t3);
            float u[4];
            u[0] = Fc;
            for (int k = 0; k < 3; ++k) u[k + 1] = (t1[k] + cr[k]) - t3[k];
            mat4_chain(c->Binv, u, Td);
        } else if (c->action_type == VFO_ACT_VELOCITY || c->action_type == VFO_ACT_POSITION) {
            geometric_controller(c, a, p, q, v, w, al, Td, vstr);
        } else {
            /* THRUST: dynamics.py:712-714,398-399 */
            for (int k = 0; k < 4; ++k) Td[k] = c->m * (a[k] * c->acc_half + c->acc_mean);
        }
        for (int k = 0; k < 4; ++k) Td[k] = clampf(Td[k], c->T_min, c->T_max); /* :501 */

        float aa[3] = { al[0], al[1], al[2] };
        /* ---- sub-steps   dynamics.py:335-367 ---- */
        for (int s = 0; s < c->interval_steps; ++s) {
            if (c->ctrl_delay) {
                for (int k = 0; k < 4; ++k) {
                    /* _compute_rotor_omega :545-553 */
                    float d1 = c->tm2 - Td[k];
                    float d2 = c->rot_4tm0 * d1;
                    float d3 = c->rot_tm1sq - d2;
                    float sq = sqrtf(d3);
                    float wd = c->rot_scale * (c->rot_neg_tm1 + sq);
                    /* first-order motor :514 */
                    wm[k] = c->c_motor * wm[k] + c->one_minus_c * wd;
                    /* _compute_thrust :530-534 */
                    float wp = wm[k] + 0.0f;
                    T[k] = (c->tm0 * (wp * wp) + c->tm1 * wm[k]) + c->tm2;
                }
            } else {
                for (int k = 0; k < 4; ++k) T[k] = Td[k]; /* :518 */
            }
            float ft[4];
            mat4_chain(c->B, T, ft); /* :339 */

            /* body-frame velocity :342, maths.py:49 */
            quat vq = { 0.0f, v[0] + 0.0f, v[1] + 0.0f, v[2] + 0.0f };
            quat vb = qmul(qmul(qconj(q), vq), q);
            float vbv[3] = { vb.x, vb.y, vb.z };
            float u[3];
            for (int k = 0; k < 3; ++k) {
                float lin = kl[k] * vbv[k];                 /* :343 */
                float qd = (kq[k] * vbv[k]) * fabsf(vbv[k]); /* :344 */
                float drag = lin + qd;                       /* :345 */
                float zf = (k == 2 ? 1.0f : 0.0f) * ft[0];   /* z * F  :347 */
                u[k] = zf - drag;
            }
            quat uq = { 0.0f, u[0], u[1], u[2] };
            quat ra = qmul(qmul(q, uq), qconj(q));
            acc[0] = ra.x / c->m + 0.0f;
            acc[1] = ra.y / c->m + 0.0f;
            acc[2] = ra.z / c->m + c->g_z;

            const float* tq = ft + 1; /* :349 */
            float dpos[3] = { v[0] + c->wind[0], v[1] + c->wind[1], v[2] + c->wind[2] }; /* maths.py:310 */

            if (c->integrator == VFO_INT_EULER) {
                float dq[4], dw[3];
                derivs(c, q, w, tq, dq, dw);
                /* maths.py:344-347 */
                for (int k = 0; k < 3; ++k) p[k] = p[k] + dpos[k] * dt;
                q.w = q.w + dq[0] * dt; q.x = q.x + dq[1] * dt;
                q.y = q.y + dq[2] * dt; q.z = q.z + dq[3] * dt;
                for (int k = 0; k < 3; ++k) v[k] = v[k] + acc[k] * dt;
                for (int k = 0; k < 3; ++k) w[k] = w[k] + dw[k] * dt;
                for (int k = 0; k < 3; ++k) aa[k] = dw[k]; /* :351 */
            } else {
                /* REPAIRED rk4: maths.py:353-386 with (i) the
                 * caller's wind passed to every stage, (ii) the four `d_* @ ks`
                 * contractions restated as explicit elementwise weighted sums
                 * ((k1*w0 + k2*w1) + k3*w2) + k4*w3, (iii) the ks-weighted
                 * d_ori_vel returned as the angular acceleration.  acc and tau
                 * stay frozen across stages as in the reference. */
                const float ks[4] = { 1.0f / 6.0f, 2.0f / 6.0f, 2.0f / 6.0f, 1.0f / 6.0f };
                const float sl[3] = { 0.5f, 0.5f, 1.0f };
                float kq4[4][4], kw[4][3], kv[4][3], kp[4][3];
                quat qc = q;
                float vc[3] = { v[0], v[1], v[2] };
                float wc[3] = { w[0], w[1], w[2] };
                for (int st = 0; st < 4; ++st) {
                    if (st != 0) {
                        /* maths.py:366-368: x + d[:, :, st-1] * slice_ts[st-1] * dt */
                        float h = sl[st - 1];
                        qc.w = q.w + kq4[st - 1][0] * h * dt;
                        qc.x = q.x + kq4[st - 1][1] * h * dt;
                        qc.y = q.y + kq4[st - 1][2] * h * dt;
                        qc.z = q.z + kq4[st - 1][3] * h * dt;
                        for (int k = 0; k < 3; ++k) vc[k] = v[k] + kv[st - 1][k] * h * dt;
                        for (int k = 0; k < 3; ++k) wc[k] = w[k] + kw[st - 1][k] * h * dt;
                    }
                    for (int k = 0; k < 3; ++k) kp[st][k] = vc[k] + c->wind[k];
                    derivs(c, qc, wc, tq, kq4[st], kw[st]);
                    for (int k = 0; k < 3; ++k) kv[st][k] = acc[k];
                }
#define WSUM(arr, k) ((((arr)[0][k] * ks[0] + (arr)[1][k] * ks[1]) + (arr)[2][k] * ks[2]) + (arr)[3][k] * ks[3])
                float dwk[3];
                for (int k = 0; k < 3; ++k) p[k] = p[k] + WSUM(kp, k) * dt;
                q.w = q.w + WSUM(kq4, 0) * dt; q.x = q.x + WSUM(kq4, 1) * dt;
                q.y = q.y + WSUM(kq4, 2) * dt; q.z = q.z + WSUM(kq4, 3) * dt;
                for (int k = 0; k < 3; ++k) v[k] = v[k] + WSUM(kv, k) * dt;
                for (int k = 0; k < 3; ++k) { dwk[k] = WSUM(kw, k); w[k] = w[k] + dwk[k] * dt; }
                for (int k = 0; k < 3; ++k) aa[k] = dwk[k];
#undef WSUM
            }
            /* normalize :367, maths.py:226-230 */
            float nn = sqrtf(((q.w * q.w + q.x * q.x) + q.y * q.y) + q.z * q.z);
            q.w = q.w / nn; q.x = q.x / nn; q.y = q.y / nn; q.z = q.z / nn;
        }
        float t = ROW(VFO_T) + c->ctrl_dt; /* :368 */

        /* _ugly_fix :374-382 */
        p[0] = clampf(p[0], -c->pos_xy_lim, c->pos_xy_lim);
        p[1] = clampf(p[1], -c->pos_xy_lim, c->pos_xy_lim);
        p[2] = clampf(p[2], c->pos_z_lo, c->pos_z_hi);
        for (int k = 0; k < 3; ++k) v[k] = clampf(v[k], -c->vel_lim, c->vel_lim);
        for (int k = 0; k < 3; ++k) w[k] = clampf(w[k], -c->omg_lim, c->omg_lim);

        for (int k = 0; k < 3; ++k) ROW(VFO_POS + k) = p[k];
        ROW(VFO_QUAT) = q.w; ROW(VFO_QUAT + 1) = q.x; ROW(VFO_QUAT + 2) = q.y; ROW(VFO_QUAT + 3) = q.z;
        for (int k = 0; k < 3; ++k) ROW(VFO_VEL + k) = v[k];
        for (int k = 0; k < 3; ++k) ROW(VFO_OMG + k) = w[k];
        for (int k = 0; k < 4; ++k) ROW(VFO_MOT + k) = wm[k];
        for (int k = 0; k < 4; ++k) ROW(VFO_THR + k) = T[k];
        for (int k = 0; k < 3; ++k) ROW(VFO_AACC + k) = aa[k];
        for (int k = 0; k < 3; ++k) ROW(VFO_ACC + k) = acc[k];
        ROW(VFO_T) = t;

        if (obs) { /* state :779-786 */
            float* o = obs + 13 * (size_t)i;
            o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
            o[3] = q.w; o[4] = q.x; o[5] = q.y; o[6] = q.z;
            for (int k = 0; k < 3; ++k) o[7 + k] = v[k] + c->wind[k];
            for (int k = 0; k < 3; ++k) o[10 + k] = w[k];
        }
#undef ROW
    }
}

void vfo_dyn_step(const vfo_consts* c, int N, float* S, float* Q, int32_t* tick,
                  const float* klin, const float* kquad,
                  const float* action, float* obs)
{
    const int D = c->delay_steps;
    const int t0 = *tick & ~VFO_TICK_VSTRIDED;
    const int slot = (D > 0 ? t0 % D : 0) | (*tick & VFO_TICK_VSTRIDED);
#pragma omp parallel if (N >= 4096)
    {
        int i0, i1;
        thread_range(N, &i0, &i1);
        dyn_step_range(c, N, S, Q, slot, klin, kquad, action, obs, i0, i1);
    }
    *tick = (D > 0 ? (t0 + 1) % D : 0) | (*tick & VFO_TICK_VSTRIDED);
}

/* ---------- Dynamics.reset ---------- */

void vfo_dyn_reset(const vfo_consts* c, int N, float* S, float* Q, int32_t* tick,
                   const int32_t* idx, int k,
                   const float* pos, const float* quat_, const float* vel, const float* omg,
                   const float* mot, const float* thr, const float* t, const float* t_rand)
{
    const int full = (idx == NULL);
    const int n = full ? N : k;
    for (int j = 0; j < n; ++j) {
        const int i = full ? j : idx[j];
#define ROW(r) S[(size_t)(r) * N + i]
        for (int d = 0; d < 3; ++d) ROW(VFO_POS + d) = pos ? pos[3 * j + d] : 0.0f;
        for (int d = 0; d < 4; ++d) ROW(VFO_QUAT + d) = quat_ ? quat_[4 * j + d] : (d == 0 ? 1.0f : 0.0f);
        for (int d = 0; d < 3; ++d) ROW(VFO_VEL + d) = vel ? vel[3 * j + d] : 0.0f;
        for (int d = 0; d < 3; ++d) ROW(VFO_OMG + d) = omg ? omg[3 * j + d] : 0.0f;
        for (int d = 0; d < 4; ++d) ROW(VFO_MOT + d) = mot ? mot[4 * j + d] : c->w_init;
        for (int d = 0; d < 4; ++d) ROW(VFO_THR + d) = thr ? thr[4 * j + d] : c->T_init;
        for (int d = 0; d < 3; ++d) ROW(VFO_AACC + d) = 0.0f;
        for (int d = 0; d < 3; ++d) ROW(VFO_ACC + d) = 0.0f;
        if (t) ROW(VFO_T) = t[j];
        else if (full || !t_rand) ROW(VFO_T) = 0.0f;
        else ROW(VFO_T) = 0.0f + t_rand[j] * 3.14f * 2.0f; /* dynamics.py:256 */
#undef ROW
        if (Q) /* dynamics.py:243,262-263 */
            for (int s = 0; s < c->delay_steps * 4; ++s) Q[(size_t)s * N + i] = 0.0f;
    }
    if (full && tick) *tick = vel ? VFO_TICK_VSTRIDED : 0;
}

/* ---------- env layer ---------- */

/* get_observation()["state"] of the task envs: HoverEnv/NavigationEnv/RacingEnv return the raw state
 * (HoverEnv.py:62-77); HoverEnv2 [(target-p)/10, q, v/10, w/10] (HoverEnv.py:136-152); NavigationEnv2
 * [target-p, q, v, w] (NavigationEnv.py:163-183). */
void vfo_env_obs(const vfo_consts* c, const vfo_env_consts* e, int N, const float* S, float* obs)
{
    for (int i = 0; i < N; ++i) {
#define ROW(r) S[(size_t)(r) * N + i]
        float* o = obs + 13 * (size_t)i;
        float p[3] = { ROW(VFO_POS), ROW(VFO_POS + 1), ROW(VFO_POS + 2) };
        float v[3] = { ROW(VFO_VEL) + c->wind[0], ROW(VFO_VEL + 1) + c->wind[1], ROW(VFO_VEL + 2) + c->wind[2] };
        for (int d = 0; d < 4; ++d) o[3 + d] = ROW(VFO_QUAT + d);
        for (int d = 0; d < 3; ++d) {
            float w = ROW(VFO_OMG + d);
            if (e->kind == VFO_ENV_HOVER2) { o[d] = (e->target[d] - p[d]) / 10.0f; o[7 + d] = v[d] / 10.0f; o[10 + d] = w / 10.0f; }
            else if (e->kind == VFO_ENV_NAV2) { o[d] = e->target[d] - p[d]; o[7 + d] = v[d]; o[10 + d] = w; }
            else { o[d] = p[d]; o[7 + d] = v[d]; o[10 + d] = w; }
        }
#undef ROW
    }
}

/* x.norm(dim=1) for 3 / 4 columns as torch's reduce kernel rounds it for the transposed (stride (1,N))
 * views the reference passes (App. B.4; probed again for 4 columns when pinning RacingEnv) */
static inline float norm3(float x, float y, float z)
{
    return sqrtf(fmaf(z, z, fmaf(y, y, x * x)));
}
static inline float norm4(float a, float b, float c_, float d)
{
    /* (N,4) view with strides (1,N) (orientation = toTensor().T): torch's column-strided reduce
     * accumulates with FMAs, like the 3-column case; a contiguous (N,4) would round separately. */
    return sqrtf(fmaf(d, d, fmaf(c_, c_, fmaf(b, b, a * a))));
}
/* (a*b).sum(dim=1) over 3 columns: separately rounded products, fp32 adds in order */
static inline float dot3_sum(const float* a, const float* b)
{
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

static void collision_point_range(const vfo_env_consts* e, int N, const float* S, vfo_env_state* es,
                                  const int32_t* idx, int j0, int j1)
{
    for (int j = j0; j < j1; ++j) {
        const int i = idx ? idx[j] : j;
        float p[3] = { S[(size_t)(VFO_POS)*N + i], S[(size_t)(VFO_POS + 1) * N + i], S[(size_t)(VFO_POS + 2) * N + i] };
        /* hstack([p - lo, hi - p]).min(dim=1)   :347-350 ; first minimum wins */
        float cand[6];
        for (int d = 0; d < 3; ++d) { cand[d] = p[d] - e->bbox_lo[d]; cand[3 + d] = e->bbox_hi[d] - p[d]; }
        int best = 0;
        for (int d = 1; d < 6; ++d) if (cand[d] < cand[best]) best = d;
        float cp[3] = { p[0], p[1], p[2] };
        cp[best % 3] = best < 3 ? e->bbox_lo[best] : e->bbox_hi[best - 3]; /* :352 */
        for (int d = 0; d < 3; ++d) es->col_point[3 * (size_t)i + d] = cp[d];
    }
}

static void collision_flags_range(const vfo_env_consts* e, int N, const float* S, vfo_env_state* es, int i0, int i1)
{
    for (int i = i0; i < i1; ++i) {
        float p[3] = { S[(size_t)(VFO_POS)*N + i], S[(size_t)(VFO_POS + 1) * N + i], S[(size_t)(VFO_POS + 2) * N + i] };
        uint8_t oob = 0;
        for (int d = 0; d < 3; ++d) oob |= (p[d] < e->bbox_lo[d]) | (p[d] > e->bbox_hi[d]); /* :361-362 */
        float vec[3];
        for (int d = 0; d < 3; ++d) vec[d] = es->col_point[3 * (size_t)i + d] - p[d]; /* :365 */
        float dis = norm3(vec[0] - 0.0f, vec[1] - 0.0f, vec[2] - 0.0f);             /* :366 */
        for (int d = 0; d < 3; ++d) es->col_vec[3 * (size_t)i + d] = vec[d];
        es->col_dis[i] = dis;
        es->is_out_bounds[i] = oob;
        es->is_collision[i] = dis < e->uav_radius;                                    /* :367 */
        es->once_collided[i] = es->once_collided[i] | es->is_collision[i];            /* :369 */
    }
}

void vfo_update_collision(const vfo_env_consts* e, int N, const float* S, vfo_env_state* es,
                          const int32_t* idx, int k)
{
    /* NB the reference recomputes vector/dis/flags for ALL agents even for an
     * indexed call (droneEnv.py:364-369); only collision_point is indexed. */
    const int n = idx ? k : N;
#pragma omp parallel if (N >= 4096)
    {
        int j0, j1, i0, i1;
        thread_range(n, &j0, &j1);
        collision_point_range(e, N, S, es, idx, j0, j1);
#pragma omp barrier
        thread_range(N, &i0, &i1);
        collision_flags_range(e, N, S, es, i0, i1);
    }
}

static float hover_like_reward(const float* p, const float* tgt, const float* q, const float* v, const float* w)
{
    /* envs/HoverEnv.py:83-94 ; strict left-to-right, python-double coefficients
     * cast to fp32 at the multiply */
    const float c1 = (float)(-0.1 * 1 / 9), c2 = (float)-0.00001, c3 = (float)-0.002;
    float r = 0.1f + norm3(p[0] - tgt[0], p[1] - tgt[1], p[2] - tgt[2]) * c1;
    r = r + norm4(q[0] - 1.0f, q[1] - 0.0f, q[2] - 0.0f, q[3] - 0.0f) * c2;
    r = r + norm3(v[0] - 0.0f, v[1] - 0.0f, v[2] - 0.0f) * c3;
    r = r + norm3(w[0] - 0.0f, w[1] - 0.0f, w[2] - 0.0f) * c3;
    return r;
}

static void env_post_step_range(const vfo_consts* c, const vfo_env_consts* e, int N, const float* S,
                                vfo_env_state* es, int i0, int i1)
{
    for (int i = i0; i < i1; ++i) {
#define ROW(r) S[(size_t)(r) * N + i]
        float p[3] = { ROW(VFO_POS), ROW(VFO_POS + 1), ROW(VFO_POS + 2) };
        float q[4] = { ROW(VFO_QUAT), ROW(VFO_QUAT + 1), ROW(VFO_QUAT + 2), ROW(VFO_QUAT + 3) };
        float v[3] = { ROW(VFO_VEL) + c->wind[0], ROW(VFO_VEL + 1) + c->wind[1], ROW(VFO_VEL + 2) + c->wind[2] };
        float w[3] = { ROW(VFO_OMG), ROW(VFO_OMG + 1), ROW(VFO_OMG + 2) };
#undef ROW
        es->step_count[i] += 1; /* droneGymEnv.py:163 */
        uint8_t success = 0, failure = 0;
        float r = 0.0f;
        if (e->kind == VFO_ENV_HOVER || e->kind == VFO_ENV_HOVER2) {
            success = 0; /* HoverEnv.py:79-80; HoverEnv2 (:97-152) only changes the observation */
            r = hover_like_reward(p, e->target, q, v, w);
        } else if (e->kind == VFO_ENV_NAV2) {
            /* NavigationEnv2.get_success/get_failure/get_reward (NavigationEnv.py:155-224): of the many
             * terms computed there only r_target_spd + r_omega + r_success reach the returned reward */
            float dp[3] = { p[0] - e->target[0], p[1] - e->target[1], p[2] - e->target[2] };
            success = norm3(dp[0], dp[1], dp[2]) <= e->success_radius;
            failure = es->is_collision[i];
            /* get_along_vertical_vector(target - p, v)  (NavigationEnv.py:16-24) */
            float base[3] = { e->target[0] - p[0], e->target[1] - p[1], e->target[2] - p[2] };
            float bn = norm3(base[0], base[1], base[2]);
            float den = bn + 1e-8f;
            float bnorm[3] = { base[0] / den, base[1] / den, base[2] / den };
            float along = dot3_sum(v, bnorm);
            float vert[3] = { v[0] - bnorm[0] * along, v[1] - bnorm[1] * along, v[2] - bnorm[2] * along };
            float away = norm3(vert[0], vert[1], vert[2]);
            float r_target_spd = (along - away * 1.0f) * 0.02f;
            float r_omega = norm3(w[0] - 0.0f, w[1] - 0.0f, w[2] - 0.0f) * -0.001f;
            float r_success = (float)(success ? 1 : 0);
            r = 0.0f + r_target_spd;
            r = r + r_omega;
            r = r + r_success;
        } else if (e->kind == VFO_ENV_NAV) {
            /* NavigationEnv.py:81-99 */
            float dp[3] = { p[0] - e->target[0], p[1] - e->target[1], p[2] - e->target[2] };
            success = norm3(dp[0], dp[1], dp[2]) <= e->success_radius;
            float tp[3] = { e->target[0] - p[0], e->target[1] - p[1], e->target[2] - p[2] };
            float t1 = dot3_sum(v, tp) / (1e-6f + norm3(tp[0], tp[1], tp[2]));
            t1 = (t1 > 10.0f ? 10.0f : t1) * 0.01f;
            /* direction = x_axis (maths.py:123-133) */
            float dir[3];
            dir[0] = 1.0f - 2.0f * (q[2] * q[2] + q[3] * q[3]);
            dir[1] = 2.0f * (q[1] * q[2] + q[3] * q[0]);
            dir[2] = 2.0f * (q[1] * q[3] - q[2] * q[0]);
            const float thrd = (float)(3.14159265358979323846 / 18.0);
            float cs = dot3_sum(dir, v) / (1e-6f + norm3(v[0], v[1], v[2])) / 1.0f;
            cs = clampf(cs, -1.0f, 1.0f);
            float ang = c->trig_mode == 1 ? vfs_acosf_cr(cs) : vfs_acosf_u10(cs);
            ang = ang < thrd ? thrd : ang;
            float t2 = (ang - thrd) * -0.01f;
            float t3 = norm4(q[0] - 1.0f, q[1] - 0.0f, q[2] - 0.0f, q[3] - 0.0f) * (float)-0.00001;
            float t4 = norm3(v[0] - 0.0f, v[1] - 0.0f, v[2] - 0.0f) * -0.002f;
            float t5 = norm3(w[0] - 0.0f, w[1] - 0.0f, w[2] - 0.0f) * -0.002f;
            float cd = es->col_dis[i];
            float t6 = 1.0f / (cd + 0.2f) * -0.01f;
            float relu1 = 1.0f - cd; relu1 = relu1 > 0.0f ? relu1 : 0.0f;
            float vv[3] = { v[0] - 0.0f, v[1] - 0.0f, v[2] - 0.0f };
            float ap = dot3_sum(es->col_vec + 3 * (size_t)i, vv) / (1e-6f + cd);
            ap = ap > 0.0f ? ap : 0.0f;
            float t7 = relu1 * ap * -0.005f;
            /* success(bool) * (max_steps - step_count)(int) * base_r * (0.2 + 0.8/(1 + 1*|v|)) */
            float sterm = (float)((int)success * (e->max_episode_steps - es->step_count[i]));
            /* python_scalar / tensor is tensor.reciprocal() * scalar in torch (two roundings) */
            float t8 = sterm * 0.1f * (0.2f + (1.0f / (1.0f + 1.0f * norm3(v[0], v[1], v[2]))) * 0.8f);
            r = 0.1f * 0.0f + t1;
            r = r + t2; r = r + t3; r = r + t4; r = r + t5; r = r + t6; r = r + t7; r = r + t8;
        } else { /* RACING: RacingEnv.py:142-148,187-215 (is_pos_reward branch) */
            int g = es->next_gate[i];
            const float* gt = e->gates[g];
            uint8_t pass = norm3(p[0] - gt[0], p[1] - gt[1], p[2] - gt[2]) <= e->success_radius;
            es->is_pass_next[i] = pass;
            g = (g + pass) % e->n_gates;
            es->next_gate[i] = g;
            es->past_gates[i] += pass;
            success = 0;
            gt = e->gates[g]; /* reward uses the UPDATED gate index */
            r = hover_like_reward(p, gt, q, v, w);
            r = r + (float)pass * 20.0f;
        }
        es->success[i] = success;
        es->failure[i] = failure;
        es->reward[i] = r;
        es->rewards[i] = es->rewards[i] + r; /* :185 */
        uint8_t ed = es->episode_done[i] | success | failure | es->is_out_bounds[i]; /* :188 */
        if (e->is_collision_reset) ed |= es->is_collision[i];                           /* :189-190 */
        es->episode_done[i] = ed;
        es->done[i] = ed | (es->step_count[i] >= e->max_episode_steps);                /* :193 */
    }
}

void vfo_env_post_step(const vfo_consts* c, const vfo_env_consts* e, int N, const float* S,
                       vfo_env_state* es)
{
#pragma omp parallel if (N >= 4096)
    {
        int i0, i1;
        thread_range(N, &i0, &i1);
        env_post_step_range(c, e, N, S, es, i0, i1);
    }
}

/* bench.py's cpu_baseline leg: `steps` consecutive env steps (dynamics interval + bbox collision + counters / reward /
 * done masks, no resets) with ONE parallel region -- agents are independent, so every thread walks its own contiguous
 * agent chunk through all the steps (no fork-join per step, the chunk stays in that core's cache).  The same range
 * functions as the per-step entry points above; `actions` holds n_actions (N,4) batches used cyclically. */
void vfo_env_run_steps(const vfo_consts* c, const vfo_env_consts* e, int N, float* S, float* Q, int32_t* tick,
                       const float* klin, const float* kquad, const float* actions, int n_actions,
                       vfo_env_state* es, int steps)
{
    const int D = c->delay_steps;
    const int tick0 = *tick & ~VFO_TICK_VSTRIDED, vstr0 = *tick & VFO_TICK_VSTRIDED;
#pragma omp parallel
    {
        int i0, i1;
        thread_range(N, &i0, &i1);
        for (int s = 0; s < steps; ++s) {
            const int slot = (D > 0 ? (tick0 + s) % D : 0) | vstr0;
            dyn_step_range(c, N, S, Q, slot, klin, kquad, actions + (size_t)(s % n_actions) * 4 * N, NULL, i0, i1);
            collision_point_range(e, N, S, es, NULL, i0, i1);
            collision_flags_range(e, N, S, es, i0, i1);
            env_post_step_range(c, e, N, S, es, i0, i1);
        }
    }
    *tick = (D > 0 ? (tick0 + steps) % D : 0) | vstr0;
}

void vfo_env_reset_attr(int N, vfo_env_state* es, const int32_t* idx, int k)
{
    (void)N;
    for (int j = 0; j < k; ++j) { /* droneGymEnv.py:387-392 */
        int i = idx[j];
        es->reward[i] = 0.0f;
        es->rewards[i] = 0.0f;
        es->done[i] = 0;
        es->episode_done[i] = 0;
        es->step_count[i] = 0;
    }
}

/* ---------- GAE ---------- */

void vfo_gae(const float* rewards, const float* values, const float* episode_starts,
             const float* last_values, const float* dones,
             float* adv, float* ret, int T, int N, double gamma_d, double lam_d)
{
    /* utils/algorithms/common.py:119-132.  SB3 holds gamma / gae_lambda as
     * python floats: `gamma * next_values` rounds gamma to fp32 at the multiply,
     * `gamma * gae_lambda` is a double product rounded once. */
    const float gamma = (float)gamma_d;
    const float gl = (float)(gamma_d * lam_d);
    for (int i = 0; i < N; ++i) {
        float last = 0.0f;
        for (int t = T - 1; t >= 0; --t) {
            float nnt, nv;
            if (t == T - 1) { nnt = 1.0f - dones[i]; nv = last_values[i]; }
            else { nnt = 1.0f - episode_starts[(size_t)(t + 1) * N + i]; nv = values[(size_t)(t + 1) * N + i]; }
            float delta = rewards[(size_t)t * N + i] + gamma * nv * nnt - values[(size_t)t * N + i];
            last = delta + gl * nnt * last;
            adv[(size_t)t * N + i] = last;
        }
        for (int t = 0; t < T; ++t)
            ret[(size_t)t * N + i] = adv[(size_t)t * N + i] + values[(size_t)t * N + i];
    }
}

/* ---------- TD-lambda returns (SHAC critic targets) ---------- */

void vfo_td_returns(const float* r, const uint8_t* done, const uint8_t* episode_done, const float* next_value,
                    float* returns, int H, int N, double gamma_d, double lamda_d)
{
    /* utils/algorithms/common.py:893-923.  python floats enter each torch op as fp32 scalars:
     * lamda*gamma is a double product rounded once; (1.0 - lamda) likewise. */
    const float gamma = (float)gamma_d, lamda = (float)lamda_d, lg = (float)(lamda_d * gamma_d);
    const float oml = (float)(1.0 - lamda_d);
    if (!episode_done) episode_done = done;
    for (int i = 0; i < N; ++i) {
        float Ai = 0.0f, lam = 1.0f;
        float Bi = next_value[(size_t)(H - 1) * N + i] * (float)(!done[(size_t)(H - 1) * N + i]);
        for (int t = H - 1; t >= 0; --t) {
            const size_t o = (size_t)t * N + i;
            const float active = (float)(!done[o]), dm = (float)(done[o] != 0), ea = (float)(!episode_done[o]);
            lam = lam * lamda * active + dm;
            Ai = active * ((lg * Ai + gamma * next_value[o]) + ((1.0f - lam) / oml) * r[o]);
            Bi = gamma * (next_value[o] * dm * ea + Bi * active) + r[o];
            returns[o] = oml * Ai + lam * Bi;
        }
    }
}

/* ---------- transcendentals (vf_sleef.h) over arrays, for tests/test_sleef_restatement.py ---------- */
void vfo_xmath(int kind, const float* a, const float* b, float* out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        switch (kind) {
        case 0: out[i] = vfs_atan2f_u10(a[i], b[i]); break;
        case 1: out[i] = vfs_sinf_u10(a[i]); break;
        case 2: out[i] = vfs_cosf_u10(a[i]); break;
        case 3: out[i] = vfs_acosf_u10(a[i]); break;
        case 4: out[i] = vfs_sinf_cr(a[i]); break;
        case 5: out[i] = vfs_cosf_cr(a[i]); break;
        case 6: out[i] = vfs_acosf_cr(a[i]); break;
        default: out[i] = vfs_atan2f_glibc(a[i], b[i]); break;
        }
    }
}
