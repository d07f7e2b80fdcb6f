/*
 * visfly_amd.h -- C-ABI of the MI355X-native VisFly hot path (libvisfly_amd.so).
 *
 * The reference (SJTU-ViSYS-team/VisFly) is pure Python/PyTorch and has no FFI layer;
 * the seams this library sits behind are the Python duck types listed in SURVEY.md 8(b).
 * Each entry point names the reference interface it replaces (path:line relative to the
 * reference root).  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (e.g. a PyTorch-ROCm
 *     tensor's data_ptr()); fp32 unless stated; SoA arrays are [rows][N] row-major.
 *   - all calls are stream-ordered and non-blocking; nothing is allocated after *_create.
 *   - return 0 on success, a negative VF_E* code otherwise; vf_last_error() gives the text.
 *     Nothing throws across the ABI.  Handles are not thread-safe (the reference is
 *     single-threaded); one handle per device.
 *   - vf_stream_t is a hipStream_t passed as an opaque pointer (NULL = default stream).
 */
#ifndef VISFLY_AMD_H
#define VISFLY_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VF_ABI_VERSION 1

typedef void* vf_stream_t;

enum { VF_OK = 0, VF_EINVAL = -1, VF_EHIP = -2, VF_ESTATE = -3 };

/* ---- state slab: wave-tile AoSoA with 16-byte granules -------------------------------
 * The reference keeps component-major (C, N) tensors (envs/base/dynamics.py:116-123).  On
 * MI355X a power-of-two row stride puts every row of an agent on the same L1/L2 set and HBM
 * channel, so the slab is tiled per wavefront instead: 64 agents x G granules, a granule
 * being 4 fp32 of ONE agent (one global_load_dwordx4 per lane, 1 KiB per wave-instruction,
 * one contiguous G KiB chunk per wave):
 *
 *     float index of (agent i, granule g, component c) =
 *         (((i / 64) * G + g) * 64 + (i % 64)) * 4 + c          G = vf_dyn_granules(cfg)
 *
 * 3-vectors sit in components 1..3 (the pure-quaternion embedding the rotation code uses);
 * component 0 of those granules is a spare slot ("s") the env layer uses.              */
#define VF_TILE 64
enum {
    VF_G_POS = 0,   /* t,  p.x, p.y, p.z                                 */
    VF_G_QUAT = 1,  /* q.w, q.x, q.y, q.z                                */
    VF_G_VEL = 2,   /* h,  v.x, v.y, v.z   (velocity without wind; h = delay-ring head, int bits) */
    VF_G_OMG = 3,   /* s,  w.x, w.y, w.z   (body rates)                  */
    VF_G_MOT = 4,   /* motor omega 0..3                                  */
    VF_G_THR = 5,   /* rotor thrusts 0..3                                */
    VF_G_AACC = 6,  /* s,  angular acceleration of the last sub-step     */
    VF_G_ACC = 7,   /* s,  linear acceleration of the last sub-step      */
    VF_G_RING = 8,  /* delay_steps granules: delayed actions (ring)      */
    /* then, if per-agent drag is enabled: (s, k_lin xyz), (s, k_quad xyz) */
    VF_G_FIXED = 8
};

enum { VF_ACT_THRUST = 0, VF_ACT_BODYRATE = 1 };   /* utils/type.py:14-18 */
enum { VF_INT_EULER = 0, VF_INT_RK4 = 1 };         /* utils/maths.py:331,353 */

/* Constants derived once on the host (envs/base/dynamics.py:26-130,562-689).
 * Matrices are row-major.  These bits are part of the parity contract. */
typedef struct vf_dyn_cfg {
    int32_t action_type;      /* VF_ACT_*                                        */
    int32_t integrator;       /* VF_INT_*                                        */
    int32_t interval_steps;   /* int(ctrl_dt/dt)                 dynamics.py:74  */
    int32_t delay_steps;      /* int(comm_delay/ctrl_dt)         dynamics.py:75  */
    int32_t ctrl_delay;       /* first-order rotor model on/off  dynamics.py:510 */
    int32_t pad0;
    float dt, ctrl_dt;
    float m;                  /* mass                                            */
    float g_z;                /* -9.81                           dynamics.py:15  */
    float J[9], Jinv[9];      /* inertia and inverse             dynamics.py:108-110 */
    float JP[9];              /* J @ BODYRATE_PID.p              dynamics.py:405 */
    float Dm[9];              /* BODYRATE_PID.d                  dynamics.py:407 */
    float B[16], Binv[16];    /* allocation matrix and inverse   dynamics.py:111-114 */
    float c_motor;            /* exp(-dt/motor_tau)              dynamics.py:581 */
    float one_minus_c;        /* 1 - c_motor                     dynamics.py:514 */
    float tm0, tm1, tm2;      /* thrust_map                      dynamics.py:579 */
    float rot_scale;          /* 1/(2*tm0)                       dynamics.py:545 */
    float rot_neg_tm1;        /* -tm1                            dynamics.py:548 */
    float rot_tm1sq;          /* tm1^2                           dynamics.py:550 */
    float rot_4tm0;           /* 4*tm0                           dynamics.py:551 */
    float T_min, T_max;       /* thrust clamp                    dynamics.py:586-593,501 */
    float acc_half, acc_mean;     /* action de-normalisation     dynamics.py:627-658 */
    float rate_half, rate_mean;   /*                             dynamics.py:635-638 */
    float k_lin[3], k_quad[3];    /* shared drag coefficients    dynamics.py:567-568 */
    float wind[3];            /* constant wind velocity          dynamics.py:135,388 */
    float pos_xy_lim, pos_z_lo, pos_z_hi, vel_lim, omg_lim; /* _ugly_fix dynamics.py:374-382 */
    float T_init, w_init;     /* hover thrust / rotor speed      dynamics.py:85-86 */
} vf_dyn_cfg;

typedef struct vf_dyn vf_dyn;

const char* vf_last_error(void);
int32_t vf_abi_version(void);

/* Dynamics.__init__ (envs/base/dynamics.py:26-130): copies cfg; no device allocation. */
int vf_dyn_create(const vf_dyn_cfg* cfg, int32_t N, int32_t per_agent_drag, vf_dyn** out);
void vf_dyn_destroy(vf_dyn* h);

/* Number of granules per agent / number of floats of the slab for N agents
 * (N is padded up to a multiple of VF_TILE; pad lanes hold inert hover states). */
int32_t vf_dyn_granules(const vf_dyn* h);
int64_t vf_dyn_slab_floats(const vf_dyn* h);

/* Bind the caller-owned device slab (what a reference Dynamics holds as its state
 * attributes, dynamics.py:116-130), vf_dyn_slab_floats() fp32, 16-byte aligned. */
int vf_dyn_bind(vf_dyn* h, float* slab);

/* Dynamics.step (envs/base/dynamics.py:319-372): one control interval, all sub-steps fused.
 *   action  (N,4) AoS in [-1,1];  state_out (N,13) AoS [p, q wxyz, v+wind, w] or NULL. */
int vf_dyn_step(vf_dyn* h, const float* action, float* state_out, vf_stream_t stream);

/* Dynamics.reset (envs/base/dynamics.py:218-269).
 *   idx == NULL: full reset, arrays hold N rows in agent order; else k indexed agents.
 *   klin/kquad (k,3) AoS per-agent drag coefficients or NULL (kept / cfg values on full reset).
 *   pos (k,3) quat (k,4) vel (k,3) omg (k,3) mot (k,4) thr (k,4): AoS or NULL for the
 *   reference defaults (zeros / identity / hover).  t (k,) or NULL; when NULL an indexed
 *   reset sets t = t_rand[j]*3.14*2 (dynamics.py:256; t_rand = uniform [0,1) draws) or 0
 *   if t_rand is NULL too.  Queue rows of reset agents are zeroed (dynamics.py:243,262). */
int vf_dyn_reset(vf_dyn* h, const int32_t* idx, int32_t k,
                 const float* pos, const float* quat, const float* vel, const float* omg,
                 const float* mot, const float* thr, const float* t, const float* t_rand,
                 const float* klin, const float* kquad, vf_stream_t stream);

/* Time the dominant kernel with HIP events on `stream`: runs `iters` back-to-back
 * vf_dyn_step launches and returns the mean device time per launch in microseconds.
 * Measurement helper for bench.py (the roofline figure), not part of the reference. */
int vf_dyn_time_steps(vf_dyn* h, const float* action, float* state_out, int32_t iters,
                      vf_stream_t stream, float* mean_us);

#ifdef __cplusplus
}
#endif
#endif
