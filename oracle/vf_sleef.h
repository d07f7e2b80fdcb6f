/*
 * vf_sleef.h -- TEST INFRASTRUCTURE (oracle side): the transcendentals of the reference path, restated.
 *
 * What torch 2.10 (this image: SLEEF 3.8, MKL 2024.2, CPU capability AVX-512) actually runs for the reference's calls
 * (envs/base/dynamics.py:421-432,437,467; utils/maths.py:244-249; envs/NavigationEnv.py:81-99), probed with
 * oracle/probe_torch_transcendentals.py on 4M random arguments each:
 *   torch.atan2            == Sleef_atan2f16_u10 (ATen/cpu/vec/vec512/vec512_float.h:264) for the vectorised body of the loop
 *                             (blocks of 2 x 16 elements); the tail of a tensor whose length is not a multiple of 32 goes
 *                             through glibc's scalar atan2f instead.
 *   torch.sin / cos / acos -> Intel MKL VML (vsSin / vsCos / vsAcos), NOT the Sleef_*_u35 / _u10 routines the
 *                             Vectorized<float> header names: the results differ from every SLEEF variant exported by
 *                             libtorch_cpu.so (u35: 22 % / 27 % of the arguments, u10: 1.9 % / 2.3 % / 8.3 %) and are
 *                             identical for contiguous and strided inputs.  MKL is closed source: there is no published
 *                             algorithm to restate.  Of the published candidates SLEEF's u10 routines are the closest
 *                             to MKL's results (a correctly rounded sin differs for 4.8 %), always within one ulp.
 *
 * This file restates SLEEF's published single-precision u10 routines (sleefsimdsp.c, FMA build: AVX2 / AVX-512 both define
 * ENABLE_FMA_SP) lane by lane -- the double-float helpers with their FMA forms, polynomial coefficients, reduction constants,
 * special cases: vfs_atan2f_u10, vfs_sinf_u10, vfs_cosf_u10, vfs_acosf_u10.  Every one is BIT-IDENTICAL to the Sleef_*f16_u10
 * symbol of libtorch_cpu.so (probe script), so atan2 is bit-identical to torch and sin / cos / acos are one ulp away from
 * torch for 2 - 8 % of the arguments; the paths that use them are held to a stated tolerance against the reference
 * (tests/_golden.py) and to the bit between this oracle and the HIP kernels, whose copy (visfly_amd/csrc/vf_xmath.hpp)
 * carries the identical text.  sin / cos reduce the argument with a branch all lanes of a vector take together (every
 * |x| < 125); only that branch is restated -- the hot path's arguments are angles in [-2 pi, 2 pi].
 *
 * Build with -ffp-contract=off: every fused operation below is an explicit fmaf().
 */
#ifndef VF_SLEEF_H
#define VF_SLEEF_H

#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float x, y; } vfs_f2;

#define VFS_INLINE static inline
#define VFS_NAN NAN
static inline float vfs_fma32(float x, float y, float z) { return fmaf(x, y, z); }
static inline float vfs_sqrt32(float x) { return sqrtf(x); }
static inline float vfs_rint32(float x) { return rintf(x); }
static inline float vfs_abs32(float x) { return fabsf(x); }
static inline int vfs_isinf32(float x) { return isinf(x); }
static inline int vfs_isnan32(float x) { return isnan(x); }
static inline uint32_t vfs_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float vfs_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline double vfs_rint64(double x) { return rint(x); }
static inline double vfs_sqrt64(double x) { return sqrt(x); }
static inline uint64_t vfs_dbits(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double vfs_double(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

/* ==== shared text: identical in oracle/vf_sleef.h and visfly_amd/csrc/vf_xmath.hpp (tests/test_sleef_restatement.py) ==== */
VFS_INLINE float vfs_mla(float x, float y, float z) { return vfs_fma32(x, y, z); }        /* x*y + z */
VFS_INLINE float vfs_fmapn(float x, float y, float z) { return vfs_fma32(x, y, -z); }     /* x*y - z */
VFS_INLINE float vfs_fmanp(float x, float y, float z) { return vfs_fma32(-x, y, z); }     /* -x*y + z */
VFS_INLINE float vfs_mulsign(float x, float y) { return vfs_float(vfs_bits(x) ^ (vfs_bits(y) & 0x80000000u)); }
VFS_INLINE int vfs_isnegzero(float x) { return vfs_bits(x) == 0x80000000u; }
VFS_INLINE vfs_f2 vfs_f2_(float x, float y) { vfs_f2 r = { x, y }; return r; }
VFS_INLINE vfs_f2 vfs_add_f_f(float x, float y) { float s = x + y; return vfs_f2_(s, (x - s) + y); }
VFS_INLINE vfs_f2 vfs_add2_f_f(float x, float y)
{
    float s = x + y, v = s - x;
    return vfs_f2_(s, (x - (s - v)) + (y - v));
}
VFS_INLINE vfs_f2 vfs_add_f2_f(vfs_f2 x, float y) { float s = x.x + y; return vfs_f2_(s, ((x.x - s) + y) + x.y); }
VFS_INLINE vfs_f2 vfs_add_f_f2(float x, vfs_f2 y) { float s = x + y.x; return vfs_f2_(s, ((x - s) + y.x) + y.y); }
VFS_INLINE vfs_f2 vfs_add2_f_f2(float x, vfs_f2 y)
{
    float s = x + y.x, v = s - x;
    return vfs_f2_(s, ((x - (s - v)) + (y.x - v)) + y.y);
}
VFS_INLINE vfs_f2 vfs_add_f2_f2(vfs_f2 x, vfs_f2 y)
{
    float s = x.x + y.x;
    return vfs_f2_(s, (((x.x - s) + y.x) + x.y) + y.y);
}
VFS_INLINE vfs_f2 vfs_sub_f2_f2(vfs_f2 x, vfs_f2 y)
{
    float s = x.x - y.x, t = x.x - s;
    t = t - y.x;
    t = t + x.y;
    return vfs_f2_(s, t - y.y);
}
VFS_INLINE vfs_f2 vfs_mul_f2_f2(vfs_f2 x, vfs_f2 y)
{
    float r0 = x.x * y.x;
    return vfs_f2_(r0, vfs_fma32(x.x, y.y, vfs_fma32(x.y, y.x, vfs_fmapn(x.x, y.x, r0))));
}
VFS_INLINE vfs_f2 vfs_mul_f2_f(vfs_f2 x, float y)
{
    float r0 = x.x * y;
    return vfs_f2_(r0, vfs_fma32(x.y, y, vfs_fmapn(x.x, y, r0)));
}
VFS_INLINE vfs_f2 vfs_mul_f_f(float x, float y) { float r0 = x * y; return vfs_f2_(r0, vfs_fmapn(x, y, r0)); }
VFS_INLINE vfs_f2 vfs_squ_f2(vfs_f2 x)
{
    float r0 = x.x * x.x;
    return vfs_f2_(r0, vfs_fma32(x.x + x.x, x.y, vfs_fmapn(x.x, x.x, r0)));
}
VFS_INLINE vfs_f2 vfs_div_f2_f2(vfs_f2 n, vfs_f2 d)
{
    float t = 1.0f / d.x;
    float q0 = n.x * t;
    float u = vfs_fmapn(t, n.x, q0);
    float v = vfs_fmanp(d.y, t, vfs_fmanp(d.x, t, 1.0f));
    return vfs_f2_(q0, vfs_fma32(q0, v, vfs_fma32(n.y, t, u)));
}
VFS_INLINE vfs_f2 vfs_rec_f(float d) { float t = 1.0f / d; return vfs_f2_(t, t * vfs_fmanp(d, t, 1.0f)); }
VFS_INLINE vfs_f2 vfs_normalize(vfs_f2 t) { float s = t.x + t.y; return vfs_f2_(s, (t.x - s) + t.y); }
VFS_INLINE vfs_f2 vfs_scale(vfs_f2 d, float s) { return vfs_f2_(d.x * s, d.y * s); }
VFS_INLINE vfs_f2 vfs_neg(vfs_f2 d) { return vfs_f2_(-d.x, -d.y); }
VFS_INLINE vfs_f2 vfs_sqrt_f(float d)
{
    float t = vfs_sqrt32(d);
    return vfs_scale(vfs_mul_f2_f2(vfs_add2_f_f2(d, vfs_mul_f_f(t, t)), vfs_rec_f(t)), 0.5f);
}

VFS_INLINE vfs_f2 vfs_atan2k_u1(vfs_f2 y, vfs_f2 x)
{
    int q = x.x < 0.0f ? -2 : 0;
    if (x.x < 0.0f) { x.x = -x.x; x.y = -x.y; }
    const int p = x.x < y.x;
    if (p) q += 1;
    vfs_f2 s = p ? vfs_neg(x) : y;
    vfs_f2 t = p ? y : x;
    s = vfs_div_f2_f2(s, t);
    t = vfs_squ_f2(s);
    t = vfs_normalize(t);
    float u = -0.00176397908944636583328247f;
    u = vfs_mla(u, t.x, 0.0107900900766253471374512f);
    u = vfs_mla(u, t.x, -0.0309564601629972457885742f);
    u = vfs_mla(u, t.x, 0.0577365085482597351074219f);
    u = vfs_mla(u, t.x, -0.0838950723409652709960938f);
    u = vfs_mla(u, t.x, 0.109463557600975036621094f);
    u = vfs_mla(u, t.x, -0.142626821994781494140625f);
    u = vfs_mla(u, t.x, 0.199983194470405578613281f);
    t = vfs_mul_f2_f2(t, vfs_add_f_f(-0.333332866430282592773438f, u * t.x));
    t = vfs_mul_f2_f2(s, vfs_add_f_f2(1.0f, t));
    t = vfs_add_f2_f2(vfs_mul_f2_f(vfs_f2_(1.5707963705062866211f, -4.3711388286737928865e-08f), (float)q), t);
    return t;
}

/* Sleef_atan2f*_u10 */
VFS_INLINE float vfs_atan2f_u10(float y, float x)
{
    const float y_in = y;
    if (vfs_abs32(x) < 2.9387372783541830947e-39f) { x = x * (float)(1 << 24); y = y * (float)(1 << 24); }
    vfs_f2 d = vfs_atan2k_u1(vfs_f2_(vfs_abs32(y), 0.0f), vfs_f2_(x, 0.0f));
    float r = d.x + d.y;
    r = vfs_mulsign(r, x);
    const float pio2 = 1.570796326794896557998981734272f;   /* (float)(M_PI/2) */
    if (vfs_isinf32(x) || x == 0.0f) r = pio2 - (vfs_isinf32(x) ? vfs_mulsign(pio2, x) : 0.0f);
    if (vfs_isinf32(y)) r = pio2 - (vfs_isinf32(x) ? vfs_mulsign(0.785398163397448278999490867136f, x) : 0.0f);
    if (y == 0.0f) r = (vfs_bits(x) & 0x80000000u) ? 3.141592653589793116f : 0.0f;
    if (vfs_isnan32(x) || vfs_isnan32(y_in)) return VFS_NAN;
    return vfs_mulsign(r, y);
}

#define VFS_PI_A2f 3.1414794921875f
#define VFS_PI_B2f 0.00011315941810607910156f
#define VFS_PI_C2f 1.9841872589410058936e-09f
#define VFS_M_1_PIf 0.318309886183790671537767526745028724f

VFS_INLINE vfs_f2 vfs_add2_f2_f(vfs_f2 x, float y)
{
    float s = x.x + y, v = s - x.x;
    float t = (x.x - (s - v)) + (y - v);
    return vfs_f2_(s, t + x.y);
}
VFS_INLINE float vfs_mul_f_f2_f2(vfs_f2 x, vfs_f2 y)   /* high part only */
{
    return vfs_fma32(x.x, y.x, vfs_fma32(x.y, y.x, x.x * y.y));
}
/* tail shared by sinf_u10 / cosf_u10: sin(s) for the reduced double-float argument s */
VFS_INLINE float vfs_sincos_tail_u10(vfs_f2 s)
{
    const vfs_f2 t = s;
    s = vfs_squ_f2(s);
    float u = 2.6083159809786593541503e-06f;
    u = vfs_mla(u, s.x, -0.0001981069071916863322258f);
    u = vfs_mla(u, s.x, 0.00833307858556509017944336f);
    const vfs_f2 x = vfs_add_f_f2(1.0f, vfs_mul_f2_f2(vfs_add_f_f(-0.166666597127914428710938f, u * s.x), s));
    return vfs_mul_f_f2_f2(t, x);
}

/* Sleef_sinf*_u10, |x| < 125 (the branch every lane of a vector takes together on the hot path: angles in [-2 pi, 2 pi]) */
VFS_INLINE float vfs_sinf_u10(float d)
{
    const float u = vfs_rint32(d * VFS_M_1_PIf);
    const int q = (int)u;
    const float v = vfs_mla(u, -VFS_PI_A2f, d);
    vfs_f2 s = vfs_add2_f_f(v, u * -VFS_PI_B2f);
    s = vfs_add_f2_f(s, u * -VFS_PI_C2f);
    float r = vfs_sincos_tail_u10(s);
    if (q & 1) r = -r;
    return vfs_isnegzero(d) ? d : r;
}

/* Sleef_cosf*_u10, |x| < 125 */
VFS_INLINE float vfs_cosf_u10(float d)
{
    const float dq = vfs_mla(vfs_rint32(vfs_mla(d, VFS_M_1_PIf, -0.5f)), 2.0f, 1.0f);
    const int q = (int)dq;
    vfs_f2 s = vfs_add2_f_f(d, dq * (-VFS_PI_A2f * 0.5f));
    s = vfs_add2_f2_f(s, dq * (-VFS_PI_B2f * 0.5f));
    s = vfs_add2_f2_f(s, dq * (-VFS_PI_C2f * 0.5f));
    float r = vfs_sincos_tail_u10(s);
    if ((q & 2) == 0) r = -r;
    return r;
}

/* Sleef_acosf*_u10 */
VFS_INLINE float vfs_acosf_u10(float d)
{
    const int o = vfs_abs32(d) < 0.5f;
    const float x2 = o ? d * d : (1.0f - vfs_abs32(d)) * 0.5f;
    vfs_f2 x = o ? vfs_f2_(vfs_abs32(d), 0.0f) : vfs_sqrt_f(x2);
    if (vfs_abs32(d) == 1.0f) x = vfs_f2_(0.0f, 0.0f);
    float u = +0.4197454825e-1f;
    u = vfs_mla(u, x2, +0.2424046025e-1f);
    u = vfs_mla(u, x2, +0.4547423869e-1f);
    u = vfs_mla(u, x2, +0.7495029271e-1f);
    u = vfs_mla(u, x2, +0.1666677296e+0f);
    u = u * (x2 * x.x);
    vfs_f2 y = vfs_sub_f2_f2(vfs_f2_(3.1415927410125732422f / 2, -8.7422776573475857731e-08f / 2),
                             vfs_add_f_f(vfs_mulsign(x.x, d), vfs_mulsign(u, d)));
    x = vfs_add_f2_f(x, u);
    if (!o) y = vfs_scale(x, 2.0f);
    if (!o && d < 0.0f) y = vfs_sub_f2_f2(vfs_f2_(3.1415927410125732422f, -8.7422776573475857731e-08f), y);
    return y.x + y.y;
}

/* ---- "cr" mode (vf_dyn_cfg.trig_mode = 1): sin / cos / acos evaluated in fp64 and rounded ONCE to fp32 -----------------------
 * What the golden-vector generator patches torch.sin / cos / acos to (x.double() -> f -> .float(), oracle/gen_golden.py:
 * the precedent is its correctly rounded sqrt): torch's own fp32 routines here are closed-source MKL
 * VML, which nothing can restate, so the reference is run with these three replaced by a fully specified definition and the
 * velocity / position controllers and NavigationEnv's view-angle term are pinned to the BIT instead of to a tolerance.
 * The fp64 evaluation restates Sun's fdlibm (k_sin.c, k_cos.c in the FreeBSD msun form, e_rem_pio2.c's medium-size path with
 * its second iteration applied unconditionally -- 118 bits of pi/2 --, e_acos.c; "Copyright (C) 1993 by Sun Microsystems,
 * Inc. All rights reserved.  Developed at SunSoft, a Sun Microsystems, Inc. business.  Permission to use, copy, modify, and
 * distribute this software is freely granted, provided that this notice is preserved."): error < 1 ulp of fp64, so the fp32
 * result equals round(torch's fp64 result) except when that fp64 value lies within ~1e-16 relative of a rounding boundary
 * (probability ~2e-9 per call; tests/test_sleef_restatement.py measures 0 mismatches in 8 M arguments).  |x| < 1e5 rad. */
VFS_INLINE double vfs_ksin64(double x, double y)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x, w = z * z;
    const double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
    const double v = z * x;
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
VFS_INLINE double vfs_kcos64(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x, w0 = z * z;
    const double r = z * (C1 + z * (C2 + z * C3)) + (w0 * w0) * (C4 + z * (C5 + z * C6));
    const double hz = 0.5 * z, w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}
/* x = n pi/2 + (y0 + y1), |y0 + y1| <= pi/4 (+ 1 ulp); returns n mod 4 */
VFS_INLINE int vfs_rem_pio2_64(double x, double* y0, double* y1)
{
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                 pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    const double fn = vfs_rint64(x * invpio2);
    const double t = x - fn * pio2_1;                  /* exact; with pio2_1t this would be fdlibm's 1st round (85 bits) */
    double w = fn * pio2_2;                            /* 2nd round: good to 118 bits (fdlibm applies it on cancellation only) */
    const double r = t - w;
    w = fn * pio2_2t - ((t - r) - w);
    *y0 = r - w;
    *y1 = (r - *y0) - w;
    return (int)((long long)fn & 3);
}
VFS_INLINE float vfs_sinf_cr(float d)
{
    double y0, y1;
    if (vfs_isinf32(d) || vfs_isnan32(d)) return VFS_NAN;
    if (d == 0.0f) return d;                                       /* keeps -0 */
    const int n = vfs_rem_pio2_64((double)d, &y0, &y1);
    const double s = vfs_ksin64(y0, y1), c = vfs_kcos64(y0, y1);
    const double v = (n & 1) ? c : s;
    return (float)((n & 2) ? -v : v);
}
VFS_INLINE float vfs_cosf_cr(float d)
{
    double y0, y1;
    if (vfs_isinf32(d) || vfs_isnan32(d)) return VFS_NAN;
    const int n = vfs_rem_pio2_64((double)d, &y0, &y1);
    const double s = vfs_ksin64(y0, y1), c = vfs_kcos64(y0, y1);
    const double v = (n & 1) ? s : c;
    return (float)(((n + 1) & 2) ? -v : v);
}
VFS_INLINE double vfs_acos_pq(double z)
{
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    return p / q;
}
VFS_INLINE float vfs_acosf_cr(float d)
{
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pi = 3.14159265358979311600e+00;
    const double x = (double)d;
    if (vfs_isnan32(d) || d > 1.0f || d < -1.0f) return VFS_NAN;
    if (d == 1.0f) return 0.0f;
    if (d == -1.0f) return (float)(pi + 2.0 * pio2_lo);
    if (d < 0.5f && d > -0.5f) {
        if (vfs_abs32(d) <= 6.938893903907228e-18f) return (float)(pio2_hi + pio2_lo);           /* |x| <= 2^-57 */
        const double r = vfs_acos_pq(x * x);
        return (float)(pio2_hi - (x - (pio2_lo - x * r)));
    }
    if (d < 0.0f) {
        const double z = (1.0 + x) * 0.5, s = vfs_sqrt64(z);
        const double w = vfs_acos_pq(z) * s - pio2_lo;
        return (float)(pi - 2.0 * (s + w));
    }
    const double z = (1.0 - x) * 0.5, s = vfs_sqrt64(z);
    const double df = vfs_double(vfs_dbits(s) & 0xffffffff00000000ull);
    const double c = (z - df * df) / (s + df);
    const double w = vfs_acos_pq(z) * s + c;
    return (float)(2.0 * (df + w));
}

/* ---- glibc 2.35 atan2f (sysdeps/ieee754/flt-32/e_atan2f.c + s_atanf.c: the fdlibm single-precision routines, "Copyright (C)
 * 1993 by Sun Microsystems, Inc. ... Permission to use, copy, modify, and distribute this software is freely granted, provided
 * that this notice is preserved"; plain separately rounded fp32 operations) --------------------------------------------------
 * torch.atan2 runs SLEEF only through its vectorised loop, i.e. for contiguous operands; STRIDED operands go element by element
 * through std::atan2 = this routine [probe: 0 mismatches in 4096 + all special values, glibc 2.35].  One call on the path sees
 * strided operands: the auto-yaw `th.atan2(velocity_horizontal[1], velocity_horizontal[0])` of the velocity action type
 * (envs/base/dynamics.py:423-427) whenever `_velocity` is the transposed view `vel.T` that Dynamics.reset stores when it is
 * given velocities (:236; every env reset is) -- the in-place integrator updates (utils/maths.py:344) and clamp() (:381) keep
 * that layout until the next full reset; after a reset() without velocities the tensor is contiguous and SLEEF runs. */
VFS_INLINE float vfs_atanf_glibc(float x)
{
    const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f, hi3 = 1.5707962513e+00f;
    const float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f, lo3 = 7.5497894159e-08f;
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const uint32_t hx = vfs_bits(x), ix = hx & 0x7fffffffu;
    const int neg = (hx >> 31) != 0;
    int id;
    if (ix >= 0x4c000000u) {                                        /* |x| >= 2^25 */
        if (ix > 0x7f800000u) return x + x;                         /* NaN */
        return neg ? -hi3 - lo3 : hi3 + lo3;
    }
    if (ix < 0x3ee00000u) {                                         /* |x| < 0.4375 */
        if (ix < 0x31000000u) return x;                             /* |x| < 2^-29 */
        id = -1;
    } else {
        x = vfs_abs32(x);
        if (ix < 0x3f980000u) {                                     /* |x| < 1.1875 */
            if (ix < 0x3f300000u) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }       /* 7/16 <= |x| < 11/16 */
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }                                /* 11/16 <= |x| < 19/16 */
        } else {
            if (ix < 0x401c0000u) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }       /* |x| < 2.4375 */
            else { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float hi = id == 0 ? hi0 : (id == 1 ? hi1 : (id == 2 ? hi2 : hi3)), lo = id == 0 ? lo0 : (id == 1 ? lo1 : (id == 2 ? lo2 : lo3));
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return neg ? -r : r;
}
VFS_INLINE float vfs_atan2f_glibc(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const uint32_t hx = vfs_bits(x), hy = vfs_bits(y), ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    if (ix > 0x7f800000u || iy > 0x7f800000u) return x + y;        /* NaN */
    if (hx == 0x3f800000u) return vfs_atanf_glibc(y);               /* x = 1 */
    const int m = (int)((hy >> 31) & 1u) | (int)((hx >> 30) & 2u);  /* 2 sign(x) + sign(y) */
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return (hy >> 31) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000u) {
        if (iy == 0x7f800000u) return m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny));
        return m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi + tiny : -pi - tiny));
    }
    if (iy == 0x7f800000u) return (hy >> 31) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = ((int)iy - (int)ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                          /* |y / x| > 2^60 */
    else if ((hx >> 31) && k < -60) z = 0.0f;                       /* |y| / x < -2^60 */
    else z = vfs_atanf_glibc(vfs_abs32(y / x));
    if (m == 0) return z;
    if (m == 1) return vfs_float(vfs_bits(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

/* ==== end of shared text ==== */

#endif /* VF_SLEEF_H */
