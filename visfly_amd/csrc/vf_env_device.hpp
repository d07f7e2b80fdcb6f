// Per-agent env arithmetic fused behind the dynamics interval (gfx950): bbox collision,
// task success / reward, counters and done masks, on-device spawning.
//
// Rounding follows the reference's torch ops: elementwise ops rounded
// separately, x.norm(dim=1) over the reference's transposed (N,3)/(N,4) views = an FMA chain
// sqrt(fma(z,z,fma(y,y,x*x))) (4 columns: one more fma), (a*b).sum(dim=1) = (a0*b0 + a1*b1) + a2*b2, python scalars cast
// to fp32 at the op, `scalar / tensor` = reciprocal(tensor) * scalar.
#pragma once
#include "vf_common.hpp"
#include "vf_dyn_device.hpp"

#pragma clang fp contract(off)

namespace vf {

__device__ __forceinline__ float norm3(float x, float y, float z)
{
    return vf_sqrt(__builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x)));
}
__device__ __forceinline__ float norm4(float a, float b, float c, float d)
{
    return vf_sqrt(__builtin_fmaf(d, d, __builtin_fmaf(c, c, __builtin_fmaf(b, b, a * a))));
}
__device__ __forceinline__ float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

struct Collision {
    float cp[3], vec[3], dis;
    bool hit, oob;
};

// DroneEnvsBase.update_collision, bbox branch (envs/base/droneEnv.py:345-369)
__device__ __forceinline__ Collision bbox_collision(const vf_env_cfg& e, const float* p)
{
    Collision c;
    float best = p[0] - e.bbox_lo[0];
    int bi = 0;
#pragma unroll
    for (int d = 1; d < 6; ++d) {  // hstack([p - lo, hi - p]).min(dim=1): first minimum wins
        const float v = d < 3 ? p[d] - e.bbox_lo[d] : e.bbox_hi[d - 3] - p[d - 3];
        if (v < best) { best = v; bi = d; }
    }
    c.oob = false;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        c.cp[d] = p[d];
        if (bi == d) c.cp[d] = e.bbox_lo[d];
        if (bi == d + 3) c.cp[d] = e.bbox_hi[d];
        c.oob = c.oob || (p[d] < e.bbox_lo[d]) || (p[d] > e.bbox_hi[d]);
        c.vec[d] = c.cp[d] - p[d];
    }
    c.dis = norm3(c.vec[0], c.vec[1], c.vec[2]);
    c.hit = c.dis < e.uav_radius;
    return c;
}

// HoverEnv.get_reward (envs/HoverEnv.py:83-94), also the positional RacingEnv reward
__device__ __forceinline__ float hover_reward(const float* p, const float* tgt, const Quat& q, const float* v,
                                              const float* w)
{
    const float c1 = (float)(-0.1 * 1 / 9), c2 = (float)-0.00001, c3 = (float)-0.002;
    float r = 0.1f + norm3(p[0] - tgt[0], p[1] - tgt[1], p[2] - tgt[2]) * c1;
    r = r + norm4(q.w - 1.0f, q.x, q.y, q.z) * c2;
    r = r + norm3(v[0], v[1], v[2]) * c3;
    r = r + norm3(w[0], w[1], w[2]) * c3;
    return r;
}

// NavigationEnv.get_reward (envs/NavigationEnv.py:84-99); step_count already incremented
__device__ __forceinline__ float nav_reward(const vf_env_cfg& e, const float* p, const Quat& q, const float* v,
                                            const float* w, const Collision& col, bool success, int step_count, int trig_mode)
{
    const float tp[3] = {e.target[0] - p[0], e.target[1] - p[1], e.target[2] - p[2]};
    float t1 = dot3(v, tp) / (1e-6f + norm3(tp[0], tp[1], tp[2]));
    t1 = (t1 > 10.0f ? 10.0f : t1) * 0.01f;
    float dir[3];  // Quaternion.x_axis (utils/maths.py:123-133)
    dir[0] = 1.0f - 2.0f * (q.y * q.y + q.z * q.z);
    dir[1] = 2.0f * (q.x * q.y + q.z * q.w);
    dir[2] = 2.0f * (q.x * q.z - q.y * q.w);
    const float thrd = (float)(3.14159265358979323846 / 18.0);
    const float vn = norm3(v[0], v[1], v[2]);
    float cs = dot3(dir, v) / (1e-6f + vn) / 1.0f;
    cs = clampf(cs, -1.0f, 1.0f);
    float ang = trig_mode == VF_TRIG_CR ? vfs_acosf_cr(cs) : vfs_acosf_u10(cs);
    ang = ang < thrd ? thrd : ang;
    const float t2 = (ang - thrd) * -0.01f;
    const float t3 = norm4(q.w - 1.0f, q.x, q.y, q.z) * (float)-0.00001;
    const float t4 = vn * -0.002f;
    const float t5 = norm3(w[0], w[1], w[2]) * -0.002f;
    const float t6 = 1.0f / (col.dis + 0.2f) * -0.01f;
    float relu1 = 1.0f - col.dis;
    relu1 = relu1 > 0.0f ? relu1 : 0.0f;
    float ap = dot3(col.vec, v) / (1e-6f + col.dis);
    ap = ap > 0.0f ? ap : 0.0f;
    const float t7 = relu1 * ap * -0.005f;
    const float sterm = (float)(success ? e.max_episode_steps - step_count : 0);
    const float t8 = sterm * 0.1f * (0.2f + (1.0f / (1.0f + 1.0f * vn)) * 0.8f);
    float r = 0.1f * 0.0f + t1;
    r = r + t2; r = r + t3; r = r + t4; r = r + t5; r = r + t6; r = r + t7; r = r + t8;
    return r;
}

// NavigationEnv2.get_reward (envs/NavigationEnv.py:185-224): of the terms computed there only
// r_target_spd + r_omega + r_success reach the returned value; get_along_vertical_vector (:16-24)
__device__ __forceinline__ float nav2_reward(const vf_env_cfg& e, const float* p, const float* v, const float* w, bool success)
{
    const float base[3] = {e.target[0] - p[0], e.target[1] - p[1], e.target[2] - p[2]};
    const float den = norm3(base[0], base[1], base[2]) + 1e-8f;
    const float bn[3] = {base[0] / den, base[1] / den, base[2] / den};
    const float along = dot3(v, bn);
    const float away = norm3(v[0] - bn[0] * along, v[1] - bn[1] * along, v[2] - bn[2] * along);
    float r = 0.0f + (along - away * 1.0f) * 0.02f;
    r = r + norm3(w[0], w[1], w[2]) * -0.001f;
    r = r + (success ? 1.0f : 0.0f);
    return r;
}

// observation variants: HoverEnv2 [(target-p)/10, q, v/10, w/10] (HoverEnv.py:136-152), NavigationEnv2
// [target-p, q, v, w] (NavigationEnv.py:163-183); o holds the raw state row [p, q, v+wind, w] on entry
__device__ __forceinline__ void obs_variant(const vf_env_cfg& e, float* o)
{
    if (e.obs_mode == VF_OBS_HOVER2) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { o[d] = (e.target[d] - o[d]) / 10.0f; o[7 + d] = o[7 + d] / 10.0f; o[10 + d] = o[10 + d] / 10.0f; }
    } else if (e.obs_mode == VF_OBS_NAV2) {
#pragma unroll
        for (int d = 0; d < 3; ++d) o[d] = e.target[d] - o[d];
    }
}

// RacingEnv2's observation inside the persistent launches (RacingEnv.py:254-262; the arithmetic of k_race_obs, vf_obs.hip): the kernels are
// instantiated with the kernel-side env kind VF_ENV_RACING2 -- RacingEnv's step (dynamics, gates, reward, re-spawn) with 16-wide rows
constexpr int VF_ENV_RACING2 = 3;
constexpr bool kind_is_racing(int kind) { return kind == VF_ENV_RACING || kind == VF_ENV_RACING2; }
constexpr int obs_width(int kind) { return kind == VF_ENV_RACING2 ? 16 : 13; }
// o: the raw state row [p, q, v, w] (13) -> out (16) for gate index `gate`
__device__ __forceinline__ void race2_obs(const vf_env_cfg& e, const float* o, int gate, float* out)
{
    const int g1 = gate + 1 == e.n_gates ? 0 : gate + 1;            // (gate + 1) % n_gates, gate < n_gates
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        out[c] = (e.gates[gate][c] - o[c]) / e.sense_radius;
        out[3 + c] = (e.gates[g1][c] - o[c]) / e.sense_radius;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) out[6 + c] = o[3 + c];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        out[10 + c] = o[7 + c] / 10.0f;
        out[13 + c] = o[10 + c] / 10.0f;
    }
}
// its adjoint (the gate index carries no gradient): d (16) -> the raw row's (13): dp = -(d[0:3] + d[3:6]) / R, dq = d[6:10],
// dv = d[10:13] / 10, dw = d[13:16] / 10 (RacingEnv2.backward_step's arithmetic, visfly_amd/envs/tasks.py)
__device__ __forceinline__ void race2_obs_bwd(const vf_env_cfg& e, const float* d, float* dd)
{
#pragma unroll
    for (int c = 0; c < 3; ++c) dd[c] = -(d[c] + d[3 + c]) / e.sense_radius;
#pragma unroll
    for (int c = 0; c < 4; ++c) dd[3 + c] = d[6 + c];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        dd[7 + c] = d[10 + c] / 10.0f;
        dd[10 + c] = d[13 + c] / 10.0f;
    }
}

// ... and their adjoint: the gradient w.r.t. an observation row in the env's obs_mode -> the gradient w.r.t. the raw state row
// (HoverEnv2 / NavigationEnv2 take requires_grad like every env of the reference: HoverEnv.py:105,125, NavigationEnv.py:109,137; r05)
__device__ __forceinline__ void obs_variant_bwd(const vf_env_cfg& e, float* d)
{
    if (e.obs_mode == VF_OBS_HOVER2) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { d[k] = -(d[k] / 10.0f); d[7 + k] = d[7 + k] / 10.0f; d[10 + k] = d[10 + k] / 10.0f; }
    } else if (e.obs_mode == VF_OBS_NAV2) {
#pragma unroll
        for (int k = 0; k < 3; ++k) d[k] = -d[k];
    }
}

// (Philox4x32-10 and u01: vf_common.hpp)


// sin and cos of one angle with ONE range reduction (Cody-Waite on pi/2, cephes-style minimax polynomials on
// [-pi/4, pi/4]; ~1 ulp for the |x| < 1e3 a spawn half-angle can take).  The spawner sits on the step's critical path
// whenever one agent of one wave ends an episode (one wave per SIMD: the launch lasts as long as its slowest wave), so its
// six libm sinf / cosf calls (~45 instructions each) were ~0.4 us of every such step.
__device__ __forceinline__ void sincos_spawn(float x, float& sn, float& cs)
{
    const float k = rintf(x * 0.636619772367581343f);
    float r = fmaf(k, -1.5707962512969971f, x);          // pi/2 split in two: hi has 9 trailing zero bits
    r = fmaf(k, -7.5497894158615964e-8f, r);
    const float z = r * r;
    const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                          fmaf(-0.5f, z, 1.0f));
    const int q = (int)k;
    const float a = (q & 1) ? pc : ps, b = (q & 1) ? ps : pc;
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}

// UniformStateRandomizer._generate + safe_generate (utils/randomization.py:64-96,153-170) and
// UnionRandomizer (:284-296) for one agent.  Draw order per agent mirrors the reference
// (pos, ori, vel, ang-vel, then the union pick and t); the stream itself is Philox keyed by
// (seed, agent, episode) instead of the reference's global MT19937, so spawns are statistically
// -- not bitwise -- equivalent; bitwise parity uses host-replayed states (vf_env_reset).
// Three Philox blocks per spawn: the twelve uniforms take the low 24 bits of the twelve words, the union pick and the
// indexed reset's t take the twelve spare top bytes (Philox output bits are independent; a fourth block was 90 instructions).
__device__ __forceinline__ void spawn_agent(const vf_env_cfg& e, int agent, unsigned episode, bool indexed, Agent& s)
{
    const unsigned k0 = (unsigned)e.seed, k1 = (unsigned)(e.seed >> 32);
    const U4 r0 = philox4x32_10(U4{(unsigned)agent, episode, 0u, 0x5eedu}, k0, k1);
    const U4 r1 = philox4x32_10(U4{(unsigned)agent, episode, 1u, 0x5eedu}, k0, k1);
    const U4 r2 = philox4x32_10(U4{(unsigned)agent, episode, 2u, 0x5eedu}, k0, k1);
    const float u[12] = {u01(r0.x), u01(r0.y), u01(r0.z), u01(r0.w), u01(r1.x), u01(r1.y),
                         u01(r1.z), u01(r1.w), u01(r2.x), u01(r2.y), u01(r2.z), u01(r2.w)};
    const unsigned pick = (r0.x >> 24) | ((r0.y >> 24) << 8) | ((r0.z >> 24) << 16) | ((r0.w >> 24) << 24);
    const unsigned tbits = (r1.x >> 24) | ((r1.y >> 24) << 8) | ((r1.z >> 24) << 16);
    int b = 0;
    if (e.n_spawn > 1) b = (int)(pick % (unsigned)e.n_spawn);  // th.randint(0, M)
    const vf_spawn_box& sb = e.spawn[b];
    float eul[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        s.p[d] = sb.pos_mean[d] + (2.0f * u[d] - 1.0f) * sb.pos_half[d];                 // :154-156
        eul[d] = (2.0f * u[3 + d] - 1.0f) * sb.ori_half[d] + sb.ori_mean[d];             // :167
        s.v[d] = (2.0f * u[6 + d] - 1.0f) * sb.vel_half[d] + sb.vel_mean[d];             // :168
        s.w[d] = (2.0f * u[9 + d] - 1.0f) * sb.omg_half[d] + sb.omg_mean[d];             // :169
    }
    // Quaternion.from_euler(roll, pitch, yaw), zyx (utils/maths.py:256-269)
    float cy, sy, cp, sp, cr, sr;
    sincos_spawn(eul[2] * 0.5f, sy, cy);
    sincos_spawn(eul[1] * 0.5f, sp, cp);
    sincos_spawn(eul[0] * 0.5f, sr, cr);
    s.q.w = cr * cp * cy + sr * sp * sy;
    s.q.x = sr * cp * cy - cr * sp * sy;
    s.q.y = cr * sp * cy + sr * cp * sy;
    s.q.z = cr * cp * sy - sr * sp * cy;
    s.t = indexed ? 0.0f + u01(tbits) * 3.14f * 2.0f : 0.0f;                             // dynamics.py:236,256
}

// Drag domain randomisation (dynamics.py:244-246): k = k_mean * (clamp((U - .5) * 2 r, -.5, .5) + 1),
// one factor per axis and per coefficient set, drawn per agent at every (re)spawn.
__device__ __forceinline__ void spawn_drag(const vf_dyn_cfg& c, const vf_env_cfg& e, int agent, unsigned episode,
                                           float4& kl, float4& kq)
{
    const unsigned k0 = (unsigned)e.seed, k1 = (unsigned)(e.seed >> 32);
    const U4 a = philox4x32_10(U4{(unsigned)agent, episode, 4u, 0x5eedu}, k0, k1);
    const U4 b = philox4x32_10(U4{(unsigned)agent, episode, 5u, 0x5eedu}, k0, k1);
    const float r2 = 2.0f * e.drag_random;
    const float ul[3] = {u01(a.x), u01(a.y), u01(a.z)}, uq[3] = {u01(b.x), u01(b.y), u01(b.z)};
    float fl[3], fq[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        fl[d] = c.k_lin[d] * (clampf((ul[d] - 0.5f) * r2, -0.5f, 0.5f) + 1.0f);
        fq[d] = c.k_quad[d] * (clampf((uq[d] - 0.5f) * r2, -0.5f, 0.5f) + 1.0f);
    }
    kl = make_float4(0.f, fl[0], fl[1], fl[2]);
    kq = make_float4(0.f, fq[0], fq[1], fq[2]);
}

// Dynamics.reset defaults for everything the spawner does not draw (dynamics.py:229-263)
__device__ __forceinline__ void reset_rotors(const vf_dyn_cfg& c, Agent& s)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) { s.wm[k] = c.w_init; s.T[k] = c.T_init; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { s.aa[k] = 0.0f; s.acc[k] = 0.0f; }
}

// RacingEnv._choose_target (envs/RacingEnv.py:172-185)
__device__ __forceinline__ int racing_choose_gate(const float* p)
{
    const float rx = p[0] - 4.0f, ry = p[1] - 0.0f;
    if (rx < 0.0f) return ry > 0.0f ? 0 : 3;
    return rx > 0.0f ? 1 : 2;
}

}  // namespace vf
