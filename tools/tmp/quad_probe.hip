// debug probe: component-layout helpers of vf_env_bwd_quad.hpp against the scalar forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "visfly_amd.h"
#include "vf_env_bwd_body.hpp"
using namespace vf;

__global__ void probe(const vf_dyn_cfg* cp, const float* in, float* out)
{
    const vf_dyn_cfg& c = *cp;
    const int lane = threadIdx.x, m = lane >> 2, k = lane & 3;
    const float* x = in + m * 32;
    const QuadLane L = quad_lane(c, lane);
    const Quat a{x[0], x[1], x[2], x[3]}, b{x[4], x[5], x[6], x[7]};
    const float v[3] = {x[8], x[9], x[10]}, w[3] = {x[11], x[12], x[13]}, lr[3] = {x[14], x[15], x[16]};
    float* o = out + lane * 32;
    // 0: qmul
    const Quat r = qmul(a, b);
    const float rc = qmul_c(L, q_sel4(k, a.w, a.x, a.y, a.z), q_sel4(k, b.w, b.x, b.y, b.z));
    o[0] = rc - q_sel4(k, r.w, r.x, r.y, r.z);
    // 1: sum4
    o[1] = q_sum4(q_sel4(k, a.w, a.x, a.y, a.z)) - (((a.w + a.x) + a.y) + a.z);
    // 2: mat3r J w
    float Jw[3];
    mat3(c.J, w[0], w[1], w[2], Jw);
    o[2] = q_mat3r(L.Jr, q_sel3(k, w)) - q_sel3(k, Jw);
    // 3: mat3T
    float t[3] = {0, 0, 0};
    mat3T_acc(c.Jinv, w, t);
    o[3] = (0.0f + q_mat3c(L.Jic, q_sel3(k, w))) - q_sel3(k, t);
    // 4: cross
    float cr[3];
    cross3(v, w, cr);
    o[4] = q_cross(q_sel3(k, v), q_sel3(k, w)) - q_sel3(k, cr);
    // 5,6: rotate_bwd
    {
        Quat lq{0.1f, 0.2f, 0.3f, 0.4f};
        float lx[3] = {0, 0, 0};
        rotate_bwd(a, v, lr, lq, lx);
        float lqc = q_sel4(k, 0.1f, 0.2f, 0.3f, 0.4f);
        const float lX = 0.0f + rotate_bwd_c(L, q_sel4(k, a.w, a.x, a.y, a.z), q_sel3(k, v), q_sel3(k, lr), lqc);
        o[5] = lqc - q_sel4(k, lq.w, lq.x, lq.y, lq.z);
        o[6] = k ? lX - q_sel3(k, lx) : 0.0f;
    }
    // 7,8: inv_rotate_bwd
    {
        Quat lq{0.1f, 0.2f, 0.3f, 0.4f};
        float lx[3] = {0.5f, 0.6f, 0.7f};
        inv_rotate_bwd(a, v, lr, lq, lx);
        float lqc = q_sel4(k, 0.1f, 0.2f, 0.3f, 0.4f);
        const float lxc = q_sel4(k, 0.0f, 0.5f, 0.6f, 0.7f) + inv_rotate_bwd_c(L, q_sel4(k, a.w, a.x, a.y, a.z), q_sel3(k, v), q_sel3(k, lr), lqc);
        o[7] = lqc - q_sel4(k, lq.w, lq.x, lq.y, lq.z);
        o[8] = k ? lxc - q_sel3(k, lx) : 0.0f;
    }
    // 9,10,11: derivs_bwd
    {
        Quat lq{0.1f, 0.2f, 0.3f, 0.4f};
        float lw[3] = {0.5f, 0.6f, 0.7f}, ltau[3] = {0, 0, 0};
        derivs_bwd(c, a, w, b, lr, lq, lw, ltau);
        float lqc = q_sel4(k, 0.1f, 0.2f, 0.3f, 0.4f), lwc = q_sel4(k, 0.0f, 0.5f, 0.6f, 0.7f), ltc = 0.0f;
        derivs_bwd_c(L, q_sel4(k, a.w, a.x, a.y, a.z), q_sel3(k, w), q_sel4(k, b.w, b.x, b.y, b.z), q_sel3(k, lr), lqc, lwc, ltc);
        o[9] = lqc - q_sel4(k, lq.w, lq.x, lq.y, lq.z);
        o[10] = k ? lwc - q_sel3(k, lw) : 0.0f;
        o[11] = k ? ltc - q_sel3(k, ltau) : 0.0f;
    }
    // 12, 13: derivs
    {
        float dq[4], dw[3];
        derivs(c, a, w, lr, dq, dw);
        float dqc, dwc;
        derivs_c<true>(L, q_sel4(k, a.w, a.x, a.y, a.z), q_sel3(k, w), q_sel4(k, 7.0f, lr[0], lr[1], lr[2]), dqc, dwc);
        o[12] = dqc - q_sel4(k, dq[0], dq[1], dq[2], dq[3]);
        o[13] = k ? dwc - q_sel3(k, dw) : 0.0f;
    }
    // 14: mat4r
    {
        const float xx[4] = {a.w, a.x, a.y, a.z};
        float y[4];
        mat4(c.B, xx, y);
        o[14] = q_mat4r(L.Br, q_sel4(k, a.w, a.x, a.y, a.z)) - q_sel4(k, y[0], y[1], y[2], y[3]);
    }

    // 15..21: one Euler sub-step of the reverse sweep (scalar transcription of env_step_bwd_agent's loop body) vs substep_bwd_c
    {
        const float dt = 0.0025f, inv_m = 1.0f / c.m;
        const Quat q = a;
        const float wm0[4] = {x[17] * 100 + 900, x[18] * 100 + 900, x[19] * 100 + 900, x[20] * 100 + 900};
        const float wd[4] = {x[21] * 100 + 900, x[22] * 100 + 900, x[23] * 100 + 900, x[24] * 100 + 900};
        const float kl[3] = {0.1f, 0.2f, 0.3f}, kq[3] = {0.01f, 0.02f, 0.03f};
        Quat lq{b.w, b.x, b.y, b.z};
        float lv[3] = {x[25], x[26], x[27]}, lw[3] = {x[28], x[29], x[30]}, lp[3] = {x[31], x[0], x[1]}, lwm[4] = {x[2], x[3], x[4], x[5]};
        float lwd[4] = {0, 0, 0, 0}, ldw_in[3] = {x[6], x[7], x[8]};
        QuadAdj qa{q_sel4(k, lq.w, lq.x, lq.y, lq.z), q_sel3(k, lv), q_sel3(k, lw), q_sel3(k, lp), q_sel4(k, lwm[0], lwm[1], lwm[2], lwm[3]), 0.0f, 0.0f, q_sel3(k, ldw_in)};
        substep_bwd_c<VF_INT_EULER, true>(c, L, q_sel4(k, q.w, q.x, q.y, q.z), q_sel3(k, v), q_sel3(k, w), q_sel4(k, wm0[0], wm0[1], wm0[2], wm0[3]),
                                          q_sel3(k, kl), q_sel3(k, kq), q_sel4(k, wd[0], wd[1], wd[2], wd[3]), 0.0f, dt, inv_m, qa);
        // scalar
        float wm1[4], Tt[4];
        for (int j = 0; j < 4; ++j) { wm1[j] = c.c_motor * wm0[j] + c.one_minus_c * wd[j]; Tt[j] = (c.tm0 * (wm1[j] * wm1[j]) + c.tm1 * wm1[j]) + c.tm2; }
        float ft[4];
        mat4(c.B, Tt, ft);
        const Quat vq{0.0f, v[0], v[1], v[2]};
        const Quat vb = qmul(qmul(qconj(q), vq), q);
        const float vbv[3] = {vb.x, vb.y, vb.z};
        float u[3];
        for (int j = 0; j < 3; ++j) u[j] = (j == 2 ? ft[0] : 0.0f) - (kl[j] * vbv[j] + (kq[j] * vbv[j]) * fabsf(vbv[j]));
        float lacc[3], ltau[3] = {0, 0, 0};
        Quat lq_in;
        float dq[4], dw[3];
        derivs(c, q, w, ft + 1, dq, dw);
        const Quat qt{q.w + dq[0] * dt, q.x + dq[1] * dt, q.y + dq[2] * dt, q.z + dq[3] * dt};
        const float nn = sqrtf(((qt.w * qt.w + qt.x * qt.x) + qt.y * qt.y) + qt.z * qt.z);
        const float rnn = 1.0f / nn;
        const Quat qn = qscale(qt, rnn);
        const float dotl = qn.w * lq.w + qn.x * lq.x + qn.y * lq.y + qn.z * lq.z;
        const Quat lqt{(lq.w - qn.w * dotl) * rnn, (lq.x - qn.x * dotl) * rnn, (lq.y - qn.y * dotl) * rnn, (lq.z - qn.z * dotl) * rnn};
        float ldw[3];
        for (int j = 0; j < 3; ++j) { ldw[j] = lw[j] * dt + ldw_in[j]; lacc[j] = lv[j] * dt; lv[j] += lp[j] * dt; ldw_in[j] = 0.0f; }
        lq_in = lqt;
        derivs_bwd(c, q, w, qscale(lqt, dt), ldw, lq_in, lw, ltau);
        float lra[3] = {lacc[0] * inv_m, lacc[1] * inv_m, lacc[2] * inv_m}, lu[3] = {0, 0, 0};
        rotate_bwd(q, u, lra, lq_in, lu);
        float lF = lu[2];
        float lvb[3];
        for (int j = 0; j < 3; ++j) lvb[j] = -lu[j] * (kl[j] + 2.0f * kq[j] * fabsf(vbv[j]));
        inv_rotate_bwd(q, v, lvb, lq_in, lv);
        const float lft[4] = {lF, ltau[0], ltau[1], ltau[2]};
        float lT[4];
        for (int j = 0; j < 4; ++j) lT[j] = c.B[j] * lft[0] + c.B[4 + j] * lft[1] + c.B[8 + j] * lft[2] + c.B[12 + j] * lft[3];
        for (int j = 0; j < 4; ++j) { const float l1 = lwm[j] + lT[j] * (2.0f * c.tm0 * wm1[j] + c.tm1); lwd[j] += c.one_minus_c * l1; lwm[j] = c.c_motor * l1; }
        lq = lq_in;

        {
            const float wm0c = q_sel4(k, wm0[0], wm0[1], wm0[2], wm0[3]), wdc = q_sel4(k, wd[0], wd[1], wd[2], wd[3]);
            const float wm1c = c.c_motor * wm0c + c.one_minus_c * wdc;
            const float Ttc = (c.tm0 * (wm1c * wm1c) + c.tm1 * wm1c) + c.tm2;
            const float ftc = q_mat4r(L.Br, Ttc);
            const float qc_ = q_sel4(k, q.w, q.x, q.y, q.z);
            const float vbc = qmul_c(L, qmul_c(L, q_conj(L, qc_), q_pure(L, q_sel3(k, v))), qc_);
            const float Fq = qb<0>(ftc); const float zf = L.k == 3 ? Fq : 0.0f;
            const float uc = zf - (q_sel3(k, kl) * vbc + (q_sel3(k, kq) * vbc) * fabsf(vbc));
            o[22] = wm1c - q_sel4(k, wm1[0], wm1[1], wm1[2], wm1[3]);
            o[23] = Ttc - q_sel4(k, Tt[0], Tt[1], Tt[2], Tt[3]);
            o[24] = ftc - q_sel4(k, ft[0], ft[1], ft[2], ft[3]);
            o[25] = k ? vbc - q_sel3(k, vbv) : 0.0f;
            o[26] = k ? uc - q_sel3(k, u) : 0.0f;
            if (lane == 15) printf("lane15: uc %g u2 %g zf %g ft0 %g ftc %g drag_c %g drag_s %g\n", uc, u[2], zf, ft[0], ftc, q_sel3(k, kl) * vbc + (q_sel3(k, kq) * vbc) * fabsf(vbc), kl[2] * vbv[2] + (kq[2] * vbv[2]) * fabsf(vbv[2]));
        }
        o[15] = qa.lq - q_sel4(k, lq.w, lq.x, lq.y, lq.z);
        o[16] = k ? qa.lv - q_sel3(k, lv) : 0.0f;
        o[17] = k ? qa.lw - q_sel3(k, lw) : 0.0f;
        o[18] = qa.lwm - q_sel4(k, lwm[0], lwm[1], lwm[2], lwm[3]);
        o[19] = qa.lwd - q_sel4(k, lwd[0], lwd[1], lwd[2], lwd[3]);
        o[20] = qa.ldw_in;
        o[21] = q_sel4(k, lq.w, lq.x, lq.y, lq.z);
    }
}

int main()
{
    vf_dyn_cfg c;
    memset(&c, 0, sizeof(c));
    c.m = 0.68f; c.c_motor = 0.9f; c.one_minus_c = 0.1f; c.tm0 = 1.5e-7f; c.tm1 = -2e-5f; c.tm2 = 0.01f;
    for (int i = 0; i < 9; ++i) { c.J[i] = 0.01f * (i + 1) + (i % 4 == 0 ? 1.0f : 0.0f); c.Jinv[i] = 0.02f * (9 - i) + (i % 4 == 0 ? 0.5f : 0.0f); }
    for (int i = 0; i < 16; ++i) c.B[i] = 0.1f * ((i * 7) % 11) - 0.4f;
    vf_dyn_cfg* dc;
    hipMalloc(&dc, sizeof(c));
    hipMemcpy(dc, &c, sizeof(c), hipMemcpyHostToDevice);
    float hin[16 * 32], *din, *dout, hout[64 * 32];
    unsigned s = 12345;
    for (int i = 0; i < 16 * 32; ++i) { s = s * 1664525u + 1013904223u; hin[i] = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
    hipMalloc(&din, sizeof(hin));
    hipMalloc(&dout, sizeof(hout));
    hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
    hipMemset(dout, 0, sizeof(hout));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dc, din, dout);
    hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
    const char* names[27] = {"qmul", "sum4", "mat3r", "mat3T", "cross", "rot_bwd lq", "rot_bwd lx", "inv_rot lq", "inv_rot lx", "derivs_bwd lq",
                             "derivs_bwd lw", "derivs_bwd ltau", "derivs dq", "derivs dw", "mat4r", "sub lq", "sub lv", "sub lw", "sub lwm", "sub lwd", "ldw_in", "(lq value)", "wm1", "Tt", "ft", "vb", "u"};
    for (int t = 0; t < 27; ++t) {
        float mx = 0;
        int at = -1;
        for (int l = 0; l < 64; ++l) if (fabsf(hout[l * 32 + t]) > mx || hout[l * 32 + t] != hout[l * 32 + t]) { mx = fabsf(hout[l * 32 + t]); at = l; }
        printf("%-18s max |diff| %.3e (lane %d)\n", names[t], mx, at);
    }
    return 0;
}
