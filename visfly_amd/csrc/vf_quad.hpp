// vf_quad.hpp -- FOUR LANES PER AGENT: component-layout arithmetic over DPP quads, shared by the forward interval of k_bptt_rollout
// (vf_dyn_quad.hpp) and the reverse sweep of k_bptt_reverse (vf_env_bwd_quad.hpp), whose waves hold 16 agents in 64 lanes.
//
// The four lanes of a QUAD (lanes 4 m .. 4 m + 3 = agent slot m) hold the four COMPONENTS of the agent's quantities:
//     quaternions      lane k = component k of (w, x, y, z)
//     3-vectors        lanes 1 .. 3 = (x, y, z), lane 0 = 0 -- i.e. the pure quaternion (0, x) the rotations are written with
//     rotor quantities lane k = rotor k;    [F; tau] = B T: lane 0 = collective thrust, lanes 1 .. 3 = torque
// so that a Hamilton product is 10 instructions instead of 28 (4 products with a quad-broadcast operand -- DPP quad_perm, an operand
// modifier, no LDS --, 3 permuted + sign-flipped copies of the other operand, 3 adds), a 3x3 / 4x4 matrix product is one fma per
// column with the lane's ROW of the matrix in registers, a cross product 5, a dot product / norm 1 + 3.
// EVERY component is computed by the same sequence of IEEE operations on the same values as in the one-lane-per-agent form
// (vf_dyn_device.hpp, vf_env_bwd_body.hpp): term order of qmul (a.w, a.x, a.y, a.z), `a - b c` as `a + (-b) c`, the fma chains of
// mat3 / mat4, the association of every sum.  The two forms therefore agree to the bit (tests/test_bptt_gpu.py compares the
// persistent launches with the launch-by-launch path bitwise); only WHERE a component lives differs.
#pragma once
#include "vf_dyn_device.hpp"

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

// LDS reads the compiler's wait-count insertion does not see.  k_bptt_reverse brings the NEXT step's record into the other half of an LDS
// area with global_load_lds while the current step reads its own half; hipcc cannot tell the halves apart and orders every LDS read it
// emits behind ALL outstanding LDS-DMA (s_waitcnt vmcnt(0) in front of the first read of a step = the HBM latency of the fetch it was
// meant to hide, 2-3 us per step).  The kernel waits for a fetch explicitly, a whole step after issuing it.  stride: 256 B = 16 float4.
typedef float vf_q4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lds_addr(const void* p)
{
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ float4 lds_read1_opaque(const float4* p)
{
    vf_q4 a;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(lds_addr(p)) : "memory");
    return make_float4(a.x, a.y, a.z, a.w);
}
__device__ __forceinline__ void lds_read2_opaque(const float4* p, float4& r0, float4& r1)      // p[0], p[16]
{
    vf_q4 a, b;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:256\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(lds_addr(p)) : "memory");
    r0 = make_float4(a.x, a.y, a.z, a.w); r1 = make_float4(b.x, b.y, b.z, b.w);
}
template <int STRIDE_BYTES>
__device__ __forceinline__ void lds_read4_opaque(const float4* p, float4& r0, float4& r1, float4& r2, float4& r3)      // p[0], p[s], p[2 s], p[3 s]
{
    vf_q4 a, b, c2, d;
    if constexpr (STRIDE_BYTES == 256)
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:256\n\tds_read_b128 %2, %4 offset:512\n\tds_read_b128 %3, %4 offset:768\n\t"
                     "s_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b), "=&v"(c2), "=&v"(d) : "v"(lds_addr(p)) : "memory");
    else
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\t"
                     "s_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b), "=&v"(c2), "=&v"(d) : "v"(lds_addr(p)) : "memory");
    r0 = make_float4(a.x, a.y, a.z, a.w); r1 = make_float4(b.x, b.y, b.z, b.w);
    r2 = make_float4(c2.x, c2.y, c2.z, c2.w); r3 = make_float4(d.x, d.y, d.z, d.w);
}

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


}  // namespace vf
