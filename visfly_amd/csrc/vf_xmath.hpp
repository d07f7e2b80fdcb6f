// vf_xmath.hpp -- device copy of the transcendentals of the reference path (gfx950).
//
// torch.atan2 on the reference's CPU path is SLEEF 3.8's atan2f_u10; torch.sin / cos / acos run Intel MKL VML (closed
// source), to which SLEEF's u10 routines are the closest published algorithms (one ulp away for 2 - 8 % of the arguments).
// The four routines below restate SLEEF's published code (sleefsimdsp.c, FMA build) in plain fp32 with explicit fma; the text
// between the "shared text" markers is IDENTICAL to the checker's copy (oracle/vf_sleef.h -- the product never includes that
// file; tests/test_sleef_restatement.py compares the two), so device and host produce the same bits.
// Call sites: geometric controller (vf_dyn_device.hpp; envs/base/dynamics.py:421-432,437,467, utils/maths.py:244-249),
// NavigationEnv view-angle term (vf_env_device.hpp, vf_env_bwd.hip; envs/NavigationEnv.py:81-99).
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>

#pragma clang fp contract(off)

struct vfs_f2 { float x, y; };

#define VFS_INLINE __device__ __forceinline__
#define VFS_NAN __builtin_nanf("")
__device__ __forceinline__ float vfs_fma32(float x, float y, float z) { return __builtin_fmaf(x, y, z); }
__device__ __forceinline__ float vfs_sqrt32(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float vfs_rint32(float x) { return __builtin_rintf(x); }
__device__ __forceinline__ float vfs_abs32(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ int vfs_isinf32(float x) { return __builtin_isinf(x); }
__device__ __forceinline__ int vfs_isnan32(float x) { return __builtin_isnan(x); }
__device__ __forceinline__ uint32_t vfs_bits(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float vfs_float(uint32_t u) { return __uint_as_float(u); }

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

/* ==== end of shared text ==== */
