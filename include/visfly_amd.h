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

#define VF_ABI_VERSION 10  /* 2: geometric-controller constants, env obs/reward modes, packed MLP weights, fused backward;
                              3: register-chain weight images (vf_mlp_layer.wr_off / wq_off, four-column pack_map),
                                 vf_mlp_backward_partial_floats;
                              4: vf_env_step_n / vf_env_graph_* (multi-step launch), vf_env_export_pose, vf_env_finish_step, vf_dyn_set_wind,
                                 vf_bptt_accumulate_checkpoint, vf_ppo_loss_cfg.old_value /
                                 clip_range_vf, vf_comm_* / vf_allreduce_grads (RCCL);
                              5: vf_dyn_ring_phase / vf_dyn_set_ring_phase / vf_env_set_ring_phase, capture guard on
                                 vf_dyn_step / vf_env_step, vf_shac_* / vf_twin_q_loss / vf_polyak_update,
                                 vf_dyn_cfg.trig_mode (was pad0), vf_env_cfg.spawn_prefetch, vf_env_out.done_list / done_count,
                                 vf_dyn_step_bwd, vf_debug_poison_lds, vf_bptt_rollout, vf_bptt_reverse, vf_ppo_rollout;
                              6: substep_tape argument of vf_bptt_rollout / vf_bptt_reverse, vf_mlp_desc.identity_mask (was pad0);
                              7: mean_rows / log_std_rows / reward_rows / ep_flag_rows of vf_bptt_rollout, log_std_rows of vf_bptt_reverse (td_policies.Actor classes);
                              8: vf_twin_q_update (fused critic step of SHAC), vf_mlp_forward_steps, vf_shac_accumulate_horizon 
                              9: vf_ppo_loss_cfg.row_index / obs_copy0 / obs_copy1 (vf_ppo_update on an indexed minibatch), vf_chain_plugin_*
                              10: vf_mlp_weight_grad_adam / vf_wgrad_tail (fold + norm + clip + Adam inside the weight-gradient launch);
                                  VF_ACTIVATION_* (vf_mlp_layer.relu is the activation kind, vf_mlp_bwd_layer.act, `act` argument of
                                  vf_linear_bwd_data / vf_linear_bwd_weight / _acc) */

/* clamp interval of the state-dependent log_std head of the reference's Actor (utils/policies/td_policies.py:31-32,241-243) */
#define VF_SAC_LOG_STD_MIN (-10.0f)
#define VF_SAC_LOG_STD_MAX 2.0f

typedef void* vf_stream_t;

enum { VF_OK = 0, VF_EINVAL = -1, VF_EHIP = -2, VF_ESTATE = -3, VF_EUNSUPPORTED = -4 };

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
    VF_G_VEL = 2,   /* h,  v.x, v.y, v.z   (velocity without wind; h = delay-ring head, int bits: the same for every agent,
                       = control steps since the last full reset mod delay_steps; the handle keeps the launch-uniform copy) */
    VF_G_OMG = 3,   /* s,  w.x, w.y, w.z   (body rates)                  */
    VF_G_MOT = 4,   /* motor omega 0..3                                  */
    VF_G_THR = 5,   /* rotor thrusts 0..3                                */
    VF_G_AACC = 6,  /* s,  angular acceleration of the last sub-step     */
    VF_G_ACC = 7,   /* s,  linear acceleration of the last sub-step      */
    VF_G_RING = 8,  /* delay_steps granules: delayed actions (ring)      */
    /* then, if per-agent drag is enabled: (s, k_lin xyz), (s, k_quad xyz) */
    VF_G_FIXED = 8
};

enum { VF_ACT_THRUST = 0, VF_ACT_BODYRATE = 1, VF_ACT_VELOCITY = 2, VF_ACT_POSITION = 3 };   /* utils/type.py:14-18 */
enum { VF_INT_EULER = 0, VF_INT_RK4 = 1 };         /* utils/maths.py:331,353 */

/* sin / cos / acos on the path.  torch's CPU fp32 routines for these three are closed-source MKL VML, which cannot be restated;
 *   VF_TRIG_CR    fp64 evaluation rounded once to fp32 -- what the golden-vector generator patches torch.sin / cos / acos to,
 *                 so that the fixtures pin these paths to the bit (default of the Python classes);
 *   VF_TRIG_SLEEF SLEEF's u10 fp32 routines (the closest published algorithm to torch's results: one ulp away for 2 - 8 % of
 *                 the arguments), no fp64 arithmetic.
 * atan2 is SLEEF's atan2f_u10 in both modes (that IS torch's atan2, bit for bit).  csrc/vf_xmath.hpp. */
enum { VF_TRIG_SLEEF = 0, VF_TRIG_CR = 1 };

/* Constants derived once on the host (envs/base/dynamics.py:26-130,562-689).
 * Matrices are row-major.  These bits are part of the parity contract. */
typedef struct vf_dyn_cfg {
    int32_t action_type;      /* VF_ACT_*                                        */
    int32_t integrator;       /* VF_INT_*                                        */
    int32_t interval_steps;   /* int(ctrl_dt/dt)                 dynamics.py:74  */
    int32_t delay_steps;      /* int(comm_delay/ctrl_dt)         dynamics.py:75  */
    int32_t ctrl_delay;       /* first-order rotor model on/off  dynamics.py:510 */
    int32_t trig_mode;        /* VF_TRIG_*: how sin / cos / acos of the velocity / position controllers (dynamics.py:432,438,
                                 467,473) and of NavigationEnv's view-angle term (NavigationEnv.py:91) are evaluated */
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
    /* velocity / position action types: geometric SO(3) controller, dynamics.py:414-496 */
    float vel_half, vel_mean; /* "velocity" range (position mode: position range) dynamics.py:660-685 */
    float yaw_half, yaw_mean; /* "yaw" range (velocity mode: both 0, dynamics.py:671) */
    float vel_p, vel_d, pos_d;/* VELOCITY_PID.p/.d, POSITION_PID.d  dynamics.py:416,433,456-457,468 */
    float Pm[9];              /* BODYRATE_PID.p                  dynamics.py:451,490 */
    float P12[9];             /* 1.2 * BODYRATE_PID.p            dynamics.py:491 */
} vf_dyn_cfg;

typedef struct vf_dyn vf_dyn;

const char* vf_last_error(void);
int32_t vf_abi_version(void);

/* Dynamics.__init__ (envs/base/dynamics.py:26-130): copies cfg to the handle and to a small device block on the CURRENT HIP
 * device (the step kernels read their constants through it); vf_dyn_destroy frees it.  Nothing else is allocated: the state
 * slab is the caller's. */
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

/* Delay-ring phase.  The comm-delay queue (dynamics.py:322-327) is a ring of delay_steps slots per agent; every agent pushes
 * once per step, so all agents share one head = control steps since the last full reset mod delay_steps.  The HANDLE keeps
 * that count on the host and passes the slot index with each launch (the per-agent head word in the slab is still written,
 * for the adjoint's tape).  Two consequences for callers that bypass the handle's bookkeeping:
 *   - a slab restored from a snapshot (or bound to a handle other than the one that stepped it) must be followed by
 *     vf_dyn_set_ring_phase / vf_env_set_ring_phase with the phase the snapshot was taken at (vf_*_ring_phase);
 *     vf_dyn_bind / vf_env_bind leave the phase alone;
 *   - a vf_dyn_step / vf_env_step launch captured into a caller's hipGraph (or torch CUDA graph) would bake ONE slot index
 *     into the graph and every replay would pop and push that same slot: with delay_steps > 1 both entry points therefore
 *     return VF_ESTATE while `stream` is capturing.  Use vf_env_graph_create (one graph per phase, checked at launch). */
int32_t vf_dyn_ring_phase(const vf_dyn* h);
int vf_dyn_set_ring_phase(vf_dyn* h, int32_t phase);

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

/* =====================================================================================
 * Env layer, visual=False: DroneEnvsBase.step + DroneGymEnvsBase.step + task reward/success
 * fused with the dynamics step into ONE launch (envs/base/droneEnv.py:114-125,345-379;
 * envs/base/droneGymEnv.py:141-218,357-423; envs/HoverEnv.py:62-94; envs/NavigationEnv.py:63-99;
 * envs/RacingEnv.py:142-215).
 *
 * Per-agent env state rides in the spare components of the dynamics granules:
 *   VF_G_OMG.c0  = _step_count (int32 bits)            droneGymEnv.py:119
 *   VF_G_AACC.c0 = _rewards running sum (fp32)         droneGymEnv.py:121
 *   VF_G_ACC.c0  = flag word (int32 bits): VF_F_* | episode counter << 8
 * RacingEnv adds one granule after the dynamics ones: (next gate, passed gates, is_pass_next, -).
 *
 * Prefetched re-spawn (vf_env_cfg.spawn_prefetch, r03).  The on-device auto-reset (examine() -> reset_agent_by_id,
 * droneGymEnv.py:339-349,420-423) draws the new state from Philox keyed by (agent, episode) -- a pure function, so it can be
 * evaluated BEFORE the episode ends.  With one wave per SIMD a launch lasts as long as its slowest wave, and one re-spawning
 * agent made its whole wave ~1.2 us late (three Philox blocks + Euler -> quaternion).  With spawn_prefetch the slab holds two
 * copies (A / B) of "the state agent i re-spawns into next": 2 x 4 granules after the racing granule,
 *   (episode tag, p) (q) (t, v) (-, w);
 * vf_env_step launches twice the blocks: the second half are HELPER blocks that look at their 64 agents' episode counter and
 * refill stale copies (tag != episode + 1) of the buffer this launch does NOT read -- they share the SIMDs with the main waves,
 * which leave three of four issue slots idle -- while a main wave that ends an episode takes the 4 granules of the OTHER buffer
 * (up to two waves per SIMD: loaded by every lane in the step's epilogue, looked at only by an ending agent; above: loaded by
 * the ending lanes) if their tag is the episode it needs, and falls back to drawing in place otherwise (first
 * episodes after a reset, episodes of length one).  Buffers alternate with the step parity, so a copy is never read and
 * written in the same launch; results are bit-identical to the in-place draw (tests/test_env_gpu.py).
 * vf_env_rollout_fused, the two-wave split kernels (<= 32 768 agents) and vf_env_finish_step always draw in place.
 * ===================================================================================== */
enum { VF_ENV_HOVER = 0, VF_ENV_NAV = 1, VF_ENV_RACING = 2 };
enum {
    VF_F_EPISODE_DONE = 1, VF_F_ONCE_COLLIDED = 2, VF_F_COLLISION = 4, VF_F_OUT_BOUNDS = 8,
    VF_F_SUCCESS = 16, VF_F_FAILURE = 32, VF_F_DONE = 64
};
enum { VF_OBS_STATE = 0, VF_OBS_HOVER2 = 1, VF_OBS_NAV2 = 2, VF_OBS_RACE2 = 3 };
enum { VF_REWARD_DEFAULT = 0, VF_REWARD_NAV2 = 1 };
enum { VF_EP_SUCCESS = 1, VF_EP_TRUNCATED = 2, VF_EP_COLLIDED = 4, VF_EP_EPISODE_DONE = 8 };
#define VF_MAX_GATES 8
#define VF_MAX_SPAWN 4

/* one UniformStateRandomizer box (utils/randomization.py:108-170); orientation = euler r,p,y */
typedef struct vf_spawn_box {
    float pos_mean[3], pos_half[3];
    float ori_mean[3], ori_half[3];
    float vel_mean[3], vel_half[3];
    float omg_mean[3], omg_half[3];
} vf_spawn_box;

typedef struct vf_env_cfg {
    int32_t kind;                 /* VF_ENV_*                                            */
    int32_t max_episode_steps;    /* droneGymEnv.py:134                                  */
    int32_t is_collision_reset;   /* droneGymEnv.py:189                                  */
    int32_t n_gates;              /* RacingEnv.py:87-93                                  */
    float bbox_lo[3], bbox_hi[3]; /* droneEnv.py:129                                     */
    float uav_radius;             /* droneEnv.py:31,367                                  */
    float success_radius;         /* HoverEnv.py:60, NavigationEnv.py:61, RacingEnv.py:98 */
    float target[3];              /* Hover / Navigation target                           */
    float gates[VF_MAX_GATES][3];
    int32_t n_spawn;              /* 1 = Uniform, >1 = Union of Uniform boxes (randomization.py:250-296) */
    float drag_random;            /* >0: per-agent drag factors 1 + clamp((U-.5)*2*r, -.5, .5) drawn at every
                                     (re)spawn (dynamics.py:244-246; needs per_agent_drag at create)   */
    vf_spawn_box spawn[VF_MAX_SPAWN];
    uint64_t seed;                /* Philox key of the on-device spawner                 */
    /* observation / reward variants */
    int32_t obs_mode;             /* VF_OBS_*: raw state | HoverEnv2 (HoverEnv.py:136-152) | NavigationEnv2 (NavigationEnv.py:163-183) |
                                     RacingEnv2 (RacingEnv.py:218-267; r06): the 16-column gate-relative row [(gate - p) / sense_radius,
                                     (next gate - p) / sense_radius, q, v / 10, w / 10] with the agent's CURRENT gate index.  The single-step
                                     launches (vf_env_step) return the raw row in this mode too -- which index the rows step() returns
                                     use is a batch-wide rule (vf_race_obs) --; the persistent launches (vf_bptt_rollout / vf_bptt_reverse
                                     / vf_ppo_rollout), whose policy reads the current gates, form and differentiate the 16 columns
                                     themselves: their observation slots, obs_final, terminal rows and g_obs are 16 wide then */
    int32_t reward_mode;          /* VF_REWARD_*: the kind's own reward | NavigationEnv2 reward + failure = is_collision (:155-224) */
    int32_t spawn_prefetch;       /* != 0: the next re-spawn state of every agent is kept ready in the slab (see "Prefetched re-spawn") */
    float sense_radius;           /* ABI 10 (was pad1): max_sense_radius of VF_OBS_RACE2 (droneGymEnv.py:69: 10) */
} vf_env_cfg;

/* Outputs of one env step; obs/reward/done are required, the rest may be NULL. */
typedef struct vf_env_out {
    float* obs;           /* (N,13) state observation AFTER auto-reset (droneGymEnv.py:209-218) */
    float* reward;        /* (N,)   reward of this step, pre-reset                              */
    uint8_t* done;        /* (N,)   done of this step, pre-reset                                */
    float* ep_return;     /* (N,)   episode reward sum, written where done   (collect_info :251) */
    int32_t* ep_length;   /* (N,)   episode length, written where done                     (:252) */
    uint8_t* ep_flags;    /* (N,)   VF_EP_* bits, written where done                  (:241-269) */
    float* terminal_obs;  /* (N,13) pre-reset observation rows, written where done        (:260) */
    int32_t* gate;        /* (N,)   RacingEnv next-gate observation after auto-reset            */
    int32_t* ep_past_gates; /* (N,) RacingEnv gates passed in the finished episode, written where done (RacingEnv.py:113-116) */
    int32_t* terminal_gate; /* (N,) RacingEnv "gate" entry of the terminal observation (pre-reset), written where done */
    /* compacted done list: the indices of the agents whose `done` this step set, in no particular order (one
     * atomic per wave), and their number.  vf_env_step zeroes *done_count before the launch; capacity N.  Optional (both or
     * neither); the multi-step entry points (vf_env_step_n, vf_env_rollout_fused, vf_env_graph_*) ignore them. */
    int32_t* done_list;
    int32_t* done_count;
} vf_env_out;

/* Dense per-agent view of the env state for the reference's properties (droneGymEnv.py:477-571,
 * droneEnv.py:424-504); every pointer optional. */
typedef struct vf_env_view {
    int32_t* step_count; float* rewards; uint8_t* flags /* VF_F_* */;
    float* col_point /* (N,3) */; float* col_vec /* (N,3) */; float* col_dis /* (N,) */;
    int32_t* gate; int32_t* past_gates;
} vf_env_view;

typedef struct vf_env vf_env;

int vf_env_create(const vf_dyn_cfg* dyn, const vf_env_cfg* env, int32_t N, int32_t per_agent_drag, vf_env** out);
void vf_env_destroy(vf_env* h);
int32_t vf_env_granules(const vf_env* h);
int64_t vf_env_slab_floats(const vf_env* h);
int vf_env_bind(vf_env* h, float* slab);
vf_dyn* vf_env_dyn(vf_env* h);   /* embedded dynamics handle sharing the slab (DroneEnvsBase.dynamics) */

/* Time-varying / per-agent wind (envs/base/dynamics.py:132-174,384-388: wind_settings given as strings are eval'ed into
 * functions of (t, previous wind) and re-evaluated by update_wind() at the top of every step; the value then holds for the
 * whole control interval and enters p' = v + wind and the velocity observation, :751-752).  The functions themselves are host
 * code (user lambdas); the step kernels take their result: N rows of 4 floats (x, y, z, unused), 16-byte aligned device
 * memory owned by the caller and rewritten by it before each step.  NULL returns to the constant vf_dyn_cfg.wind.  Works on
 * the embedded handle of an env (vf_env_dyn) for vf_env_step / vf_env_finish_step / vf_env_export_pose; the adjoint kernel and
 * the multi-step entry points (vf_env_step_n, vf_env_rollout_fused, vf_env_graph_*) see one wind for all their steps. */
int vf_dyn_set_wind(vf_dyn* h, const float* wind_Nx4);

/* DroneGymEnvsBase.reset / reset_agent_by_id (droneGymEnv.py:302-349, droneEnv.py:260-288).
 *   idx NULL: all agents.  full_state (k,22) = [p q v w motor thrust t] (dynamics.py:793-803)
 *   gives the spawn states (host-replayed randomizer, parity mode); NULL draws them on the
 *   device from the cfg spawn boxes with Philox4x32-10 keyed by (seed, agent, episode#).
 *   Clears counters/flags of the reset agents, recomputes their collision flags. */
int vf_env_reset(vf_env* h, const int32_t* idx, int32_t k, const float* full_state, vf_stream_t stream);

/* DroneGymEnvsBase.step (droneGymEnv.py:141-218): dynamics interval + bbox collision + counters +
 * success/failure + reward + done masks, and, if auto_reset, the examine()/reset_agent_by_id of
 * done agents with on-device spawning.  action (N,4) AoS in [-1,1]. */
int vf_env_step(vf_env* h, const float* action, const vf_env_out* out, int32_t auto_reset, vf_stream_t stream);

/* K consecutive DroneGymEnvsBase.step calls (droneGymEnv.py:141-218) from ONE host call: the launch loop the
 * reference drives from Python (`for _ in range(n): env.step(a)`, e.g. exps/ and the test harness
 * utils/evaluate.py:62-103 with a fixed action sequence) runs in C, so that the host cost per step stays well
 * below the ~11 us kernel.  Step k reads actions + k*action_stride and writes obs + k*obs_stride,
 * reward + k*reward_stride, done + k*done_stride (strides in ELEMENTS of the respective array; a stride of 0
 * re-uses the same rows every step, e.g. one constant action, or outputs that only keep the last step).
 * The episode outputs of `out` (ep_return ... terminal_gate) are shared by all K steps: each holds the most
 * recently finished episode per agent.  Results are bit-identical to K vf_env_step calls. */
typedef struct vf_env_rollout {
    const float* actions;      /* step 0: (N,4) AoS */
    int64_t action_stride;
    vf_env_out out;            /* pointers of step 0 */
    int64_t obs_stride, reward_stride, done_stride;
    int32_t K, auto_reset;
} vf_env_rollout;
int vf_env_step_n(vf_env* h, const vf_env_rollout* r, vf_stream_t stream);

/* The same K steps in ONE launch: every thread keeps its agent in registers from step to step (the state slab is read once
 * before step 0 and written once after step K-1; per step only the action row is read and obs / reward / done / episode
 * outputs are written), which removes K-1 launch boundaries and 2(K-1) state round trips.  Results are bit-identical to K
 * vf_env_step calls, auto-resets included.  Open-loop use (the actions of all K steps exist up front: evaluation with a
 * fixed sequence, sampling-based planners, system identification); a policy in the loop needs vf_env_step.
 * action_stride must be a multiple of 4 floats. */
int vf_env_rollout_fused(vf_env* h, const vf_env_rollout* r, vf_stream_t stream);

/* The same K launches captured once into a hipGraph and replayed with one hipGraphLaunch per rollout.  The rollout's
 * pointers and the delay-ring slot of every captured launch are baked into the graph: the caller keeps those buffers alive,
 * refills `actions` between replays, and replays a graph only at the ring phase it was captured at (vf_env_ring_phase ==
 * steps since the last full reset mod delay_steps; vf_env_graph_launch returns VF_ESTATE otherwise) -- with K a multiple of
 * delay_steps one graph serves every replay, else keep one graph per phase. */
typedef struct vf_env_graph vf_env_graph;
int32_t vf_env_ring_phase(const vf_env* h);
int vf_env_set_ring_phase(vf_env* h, int32_t phase);   /* see vf_dyn_set_ring_phase */
int vf_env_graph_create(vf_env* h, const vf_env_rollout* r, vf_env_graph** out);
int vf_env_graph_launch(vf_env_graph* g, vf_stream_t stream);
void vf_env_graph_destroy(vf_env_graph* g);

/* Pose hand-off to an external renderer / scene manager: what DroneEnvsBase.step passes to
 * sceneManager.set_pose(position, orientation, velocity) when visual=True (envs/base/droneEnv.py:375-377,
 * utils/SceneManager.py:336-361) as AoS device arrays of the CURRENT state: pos (N,3), quat (N,4) wxyz
 * (16-byte aligned), vel (N,3) incl. wind (dynamics.py:751-752), omg (N,3); every pointer optional.  A renderer
 * that can read the slab layout above may instead take zero-copy views of the VF_G_POS / VF_G_QUAT granules. */
int vf_env_export_pose(vf_env* h, float* pos, float* quat, float* vel, float* omg, vf_stream_t stream);

/* The step split around an external scene manager, as DroneEnvsBase.step does it with visual=True (droneEnv.py:374-379):
 *   vf_dyn_step(vf_env_dyn(h), action, NULL, stream)           dynamics.step(action)                       :375
 *   vf_env_export_pose(h, pos, quat, vel, NULL, stream)        sceneManager.set_pose(...); .step()         :376-378
 *   ... the scene manager computes, for these poses, its closest scene point per agent and its out-of-bounds flags ...
 *   vf_env_finish_step(h, closest_point, out_bounds, out, auto_reset, stream)
 *                                                              update_collision (visual branch :330-342: collision_point from
 *                                                              the scene, vector / distance / is_collision :364-367), then the
 *                                                              rest of DroneGymEnvsBase.step (droneGymEnv.py:161-218)
 * ext_collision_point (N,3) and ext_out_bounds (N, 0/1 bytes) are device arrays; either may be NULL = this library's bounding-box
 * query, and with both NULL the two launches equal one vf_env_step bit for bit.  Agents that auto-reset inside the call take
 * the bounding-box query for their re-spawn position (the reference asks the scene manager again, :339-341). */
int vf_env_finish_step(vf_env* h, const float* ext_collision_point, const uint8_t* ext_out_bounds, const vf_env_out* out,
                       int32_t auto_reset, vf_stream_t stream);

int vf_env_query(vf_env* h, const vf_env_view* view, vf_stream_t stream);

/* RacingEnv2's observation (envs/RacingEnv.py:218-267) from the step kernel's raw state rows: state (N, 3 n_next + 10) =
 * [(gates[(g + k) % n_gates] - p) / radius for k < n_next, q, v / 10, w / 10], gate_out (N,) = g (may be NULL).  Which index g the
 * rows that step() RETURNS use depends on the whole batch (droneGymEnv.py:161-166,347: the observation is refreshed before the gate of
 * an agent that just passed one advances -- unless some agent ended its episode in the step, which rebuilds every observation with the
 * advanced gates): mode 0: g = gate[i]; 1: g = gate_prev[i] (the index at the start of the step); 2: gate[i] if *done_count > 0
 * (vf_env_out.done_count of the step just launched on the same stream) else gate_prev[i].  gates: HOST array (n_gates, 3). */
int vf_race_obs(const float* raw, const int32_t* gate, const int32_t* gate_prev, const int32_t* done_count, int32_t mode,
                const float* gates, int32_t n_gates, int32_t n_next, float radius, float* state, int32_t* gate_out, int32_t N,
                vf_stream_t stream);

/* mean device microseconds per vf_env_step launch over `iters` back-to-back launches (HIP events) */
int vf_env_time_steps(vf_env* h, const float* action, const vf_env_out* out, int32_t auto_reset, int32_t iters,
                      vf_stream_t stream, float* mean_us);

/* Adjoint of one env step for first-order (BPTT) policy optimisation: replaces the torch autograd
 * tape over Dynamics.step + reward that `requires_grad=True` builds in the reference
 * (utils/algorithms/BPTT.py:107-134; envs/base/droneGymEnv.py:209-213; envs/base/dynamics.py:176-190).
 * The caller checkpoints the slab before each forward step (tape) and walks the steps in reverse:
 *   tape_slab    slab copy taken BEFORE the forward step (vf_env_slab_floats() floats)
 *   action       (N,4) action that was passed to that step
 *   d_obs        (N,13) dLoss/d(state observation returned by that step) or NULL
 *   d_reward     (N,)   dLoss/d(reward returned by that step) or NULL
 *   done         (N,)   done flags that step returned (reset agents stop the gradient, like the
 *                       in-place reset writes of the reference)
 *   adj_slab     in/out adjoint of the persistent state, same layout as the slab: on entry the
 *                adjoint w.r.t. the state AFTER the step, on exit w.r.t. the state BEFORE it
 *   d_action     (N,4) out: dLoss/d(action)
 * Euler and (repaired) RK4 integrators, thrust / bodyrate actions, Hover / Racing (hover-style terms) and
 * NavigationEnv rewards; anything else returns VF_EINVAL.  With per-agent wind rows set (vf_dyn_set_wind) the call returns
 * VF_EUNSUPPORTED: the tape does not record the rows, and replaying the interval with the constant wind would differentiate a
 * different trajectory. */
typedef struct vf_env_bwd_args {
    const float* tape_slab; const float* action; const float* d_obs; const float* d_reward;
    const uint8_t* done; float* adj_slab; float* d_action;
} vf_env_bwd_args;
int vf_env_step_bwd(vf_env* h, const vf_env_bwd_args* args, vf_stream_t stream);
/* Dynamics.step's own reverse pass: the same kernel on a bare vf_dyn handle -- no reward term, no episode ends.
 *   tape_slab copy of the slab taken BEFORE the vf_dyn_step call, action what that call was given, d_state (N,13) dLoss/d(state it
 *   returned) or NULL, adj_slab in/out as above, d_action (N,4) out.  Thrust / bodyrate actions, Euler / RK4. */
int vf_dyn_step_bwd(vf_dyn* h, const float* tape_slab, const float* action, const float* d_state, float* adj_slab,
                    float* d_action, vf_stream_t stream);
/* SURVEY 8b.4 also lists `*_cpu` twins of these entry points in the CPU restatement library.  The restatement (oracle/
 * vf_oracle.h, test infrastructure) keeps an ABI of its own instead -- vfo_dyn_step, vfo_env_post_step, vfo_gae, vfo_td_returns ...
 * on the reference's (C, N) row layout with host pointers -- because it restates the REFERENCE's data layout and call
 * structure (one function per reference method), not this library's fused launches; the parity tests drive both through their
 * Python wrappers (oracle.OracleDynamics / OracleEnv vs visfly_amd.Dynamics / envs). */

/* =====================================================================================
 * PPO inner loop on the device (utils/algorithms/PPO.py:177-337; SB3 2.2.1 RolloutBuffer /
 * collect_rollouts, mirrored in utils/algorithms/common.py:97-132; utils/policies/policies.py:195-254;
 * utils/policies/extractors.py:376-449).  All matrices row-major fp32; GEMMs run on the fp32 MFMA
 * (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered FMA chain), fp32 accumulate.
 * ===================================================================================== */

/* GAE backward scan, thread per env (common.py:119-132).  [T][N] arrays; gamma/lam are the python
 * floats SB3 holds (gamma rounded to fp32 at the multiply, gamma*lam rounded once). */
int vf_gae(const float* rewards, const float* values, const float* episode_starts, const float* last_values,
           const float* dones, float* adv, float* ret, int32_t T, int32_t N, double gamma, double lam,
           vf_stream_t stream);

/* TD-lambda returns with a per-env lambda reset on done (SHAC critic targets,
 * utils/algorithms/common.py:893-923), thread per env.  [H][N] arrays; episode_done NULL = done. */
int vf_td_returns(const float* r, const uint8_t* done, const uint8_t* episode_done, const float* next_value,
                  float* returns, int32_t H, int32_t N, double gamma, double lamda, vf_stream_t stream);

/* Advantage normalisation of one minibatch (PPO.py:215-220): (A - mean) / (std_unbiased + 1e-8).
 * scratch: >= 2*1024 floats.  If sums_inout != NULL the two partial sums (sum, sum of squares, fp64
 * as 2 doubles) are left there after `phase` 0 and consumed in `phase` 1, so that a multi-GPU caller
 * can all-reduce them in between (count = global element count); phase 2 = both in one call. */
int vf_adv_normalize(const float* adv, float* out, int64_t n, int64_t count, double* sums_inout, float* scratch,
                     int32_t phase, vf_stream_t stream);

/* The same normalisation for n_seg consecutive minibatches of seg_len rows in one launch (the epoch buffer
 * is shuffled once, so minibatches are contiguous slices).  sums: n_seg x 2 doubles (sum, sum of squares),
 * written by phase 0 / 2, read by phase 1 / 2 (all-reduce them in between for a multi-GPU minibatch);
 * count = rows of the GLOBAL minibatch. */
int vf_adv_normalize_segments(const float* adv, float* out, int32_t n_seg, int64_t seg_len, int64_t count, double* sums,
                              int32_t phase, vf_stream_t stream);

/* Activation of an MLP layer (ABI 10): create_mlp's `activation_fn` (utils/policies/extractors.py:376-449) with the aliases of
 * CustomMultiInputActorCriticPolicy (utils/policies/policies.py:64-69: relu, tanh, elu, leaky_relu -- torch.nn defaults: ELU alpha 1,
 * LeakyReLU slope 0.01).  The values 0 / 1 are the `relu` flag of ABI <= 9.  In all four the derivative is a function of the layer's
 * OUTPUT, so the reverse sweep needs the saved output only: [y > 0] | 1 - y^2 | y > 0 ? 1 : y + 1 | y > 0 ? 1 : 0.01.
 * The register-chained kernels built into the library are ReLU networks; other activations run on the block-tile kernels or on chain
 * classes generated for them (visfly_amd/_jit.py). */
#define VF_ACTIVATION_NONE 0
#define VF_ACTIVATION_RELU 1
#define VF_ACTIVATION_TANH 2
#define VF_ACTIVATION_ELU 3
#define VF_ACTIVATION_LEAKY_RELU 4

/* Y[M][No] (+)= act(X[M][K] @ W^T + b)   nn.Linear + activation (extractors.py:421-445)
 *   W [No][K], b [No] or NULL; ldx / ldy row strides in floats; relu: VF_ACTIVATION_*; K, No <= 128. */
int vf_linear_fwd(const float* X, int32_t ldx, const float* W, const float* b, float* Y, int32_t ldy,
                  int32_t M, int32_t K, int32_t No, int32_t relu, vf_stream_t stream);

/* dX[M][K] (+)= (dY * act'(Y)) @ W ; Ymask = the layer's saved post-activation output or NULL (no activation), act = its
 * VF_ACTIVATION_* (0 with a Ymask: ReLU); accumulate != 0 adds into dX (a tensor consumed by two branches). */
int vf_linear_bwd_data(const float* dY, int32_t lddy, const float* Ymask, int32_t ldym, const float* W,
                       float* dX, int32_t lddx, int32_t M, int32_t K, int32_t No, int32_t accumulate, int32_t act,
                       vf_stream_t stream);

/* dW[No][K] = (dY * act'(Y))^T @ X, db[No] = column sums; deterministic two-stage reduction.
 * scratch: vf_linear_bwd_scratch_floats(M, K, No) floats. */
int64_t vf_linear_bwd_scratch_floats(int32_t M, int32_t K, int32_t No);
int vf_linear_bwd_weight(const float* dY, int32_t lddy, const float* Ymask, int32_t ldym, const float* X,
                         int32_t ldx, float* dW, float* db, int32_t M, int32_t K, int32_t No, float* scratch,
                         int32_t act, vf_stream_t stream);
/* same, adding into dW / db (gradient accumulation over the steps of a BPTT horizon) */
int vf_linear_bwd_weight_acc(const float* dY, int32_t lddy, const float* Ymask, int32_t ldym, const float* X,
                             int32_t ldx, float* dW, float* db, int32_t M, int32_t K, int32_t No, float* scratch,
                             int32_t act, vf_stream_t stream);

/* Whole actor-critic MLP forward in ONE launch (policies.py:195-254: extract_features -> mlp_extractor ->
 * action_net / value_net).  A workgroup walks 64-row tiles; activations live in LDS between layers,
 * the weights come from the packed copy below as the MFMA B operand (two workgroups per CU), fp32 MFMA.
 * Buffers are numbered: 0..3 = the observation inputs (global, row-major, width in_dim[i]);
 * 4.. = LDS activation regions laid out by the host (offset / row stride in floats, stride odd);
 * VF_MLP_OUT0 / VF_MLP_OUT1 = the global outputs (row-major (M, No) of the layer that writes them: mean (M,4) and value (M,1)
 * for the actor-critic policy, mu / log_std (M,4) for the SHAC actor, Q1 / Q2 (M,1) for the twin critic).
 * A layer whose K is not a multiple of 16 and whose source is an LDS buffer (a concatenation such as features (+) action,
 * td_policies.py:137) must read the TAIL of that buffer (src_col + K = its width): the kernel zero-fills the columns
 * [src_col + K, src_col + round16(K)) of the LDS copy, which the 16-step MFMA sweep reads against zero weights. */
#define VF_MLP_MAX_LAYERS 16
#define VF_MLP_MAX_BUFS 16
enum { VF_MLP_OUT0 = 100, VF_MLP_OUT1 = 101 };
typedef struct vf_mlp_layer {
    int32_t K, No, relu;
    int32_t src, src_col;        /* buffer id and first column read  */
    int32_t dst, dst_col;        /* buffer id and first column written */
    int32_t w_off, b_off;        /* offsets into the flat parameter buffer */
    int32_t save_ld;             /* row stride of `save`, 0 if none */
    int32_t wt_off;              /* offset of this layer's packed forward weights (transposed, zero padded), see below */
    int32_t wb_off;              /* offset of this layer's packed data-gradient weights (zero padded), see below */
    int32_t wr_off;              /* offset of this layer's register-chain image (multiple of 4), see below */
    int32_t wq_off;              /* offset of this layer's reverse-chain (data-gradient) image, see below */
    float* save;                 /* optional global copy of the layer output (training keeps activations) */
} vf_mlp_layer;
typedef struct vf_mlp_desc {
    int32_t n_layers, n_inputs;
    int32_t in_dim[4];
    int32_t lds_off[VF_MLP_MAX_BUFS], lds_stride[VF_MLP_MAX_BUFS];   /* indexed by buffer id (ids < 4: staged inputs) */
    int32_t lds_floats;          /* total dynamic LDS, floats (activations only; <= 80 KiB lets two workgroups share a CU) */
    int32_t identity_mask;       /* bit i: layer i is a frozen identity (W = I, b = 0, no ReLU) that appends an input's columns to a
                                  * concatenation -- th.cat([features, actions]) of ContinuousCritic.forward (td_policies.py:137).  The
                                  * block-tile kernels run it like any layer; the register-chained classes with a pass-through input
                                  * (vf_mlp_chain_sac.hip) copy the columns instead and REQUIRE the bit, so that a trainable layer of
                                  * the same shape is never mistaken for one.  0 = none (was pad0 before ABI 6) */
    vf_mlp_layer layer[VF_MLP_MAX_LAYERS];
} vf_mlp_desc;
/* The forward reads its MFMA B operand straight from global memory: `packed` holds, per layer at float offset
 * wt_off, Wt[k][n] = W[n][k] for k < round16(K), n < round32(No), zero padded (so neither the kernel nor
 * the L1/L2-resident loads need guards); the data gradient of vf_mlp_backward reads, at wb_off,
 * Wb[n][k] = W[n][k] for n < round16(No), k < round32(K).
 * Register-chain image (wr_off): when the layer table is one of the network classes the register-chained forward is
 * instantiated for (reference default policies: one or two [128, 64] extractor branches, [64, 64] trunks), a wave64
 * carries 32 rows through the whole network with the activations in MFMA accumulator registers; its A operand is
 * read as one float4 per lane from blocks of 256 floats: block (a, g) of a layer with G = Kp / 8 reduction groups
 * (Kp = round8(K) if the layer reads an observation, else round32(K)) sits at wr_off + (a G + g) 256, and lane l,
 * word j of it holds W[32 a + (l & 31)][k], k = 8 g + 2 j + (l >> 5) for observation layers,
 * k = 32 (g / 4) + 8 (g % 4) + 4 (l >> 5) + j otherwise (the order in which an accumulator lane holds its row's
 * features), zero padded; ceil(No / 32) G 256 floats per layer.
 * Reverse-chain image (wq_off): the same construction for the data gradient dA^T = W^T dZ^T: block (a, g),
 * a < ceil(K / 32), g < G = ceil(No / 8), at wq_off + (a G + g) 256; lane l, word j holds
 * W[32 (g / 4) + 8 (g % 4) + 4 (l >> 5) + j][32 a + (l & 31)], zero padded.
 * vf_mlp_pack_weights refreshes all four images from `params` (call it after every optimiser step, or let
 * vf_adam_step do it through vf_adam_cfg.pack_map); vf_mlp_packed_floats = size of the packed buffer the layer
 * table implies. */
int64_t vf_mlp_packed_floats(const vf_mlp_desc* desc);
/* Register-chained kernels for layer tables the library holds no instance of (any depth / width of the reference's
 * `features_extractor_kwargs.net_arch` and `net_arch=dict(pi=.., vf=..)`: utils/policies/extractors.py:376-449, widths in multiples of 32
 * up to 128): the host side generates a translation unit that names the shape (csrc/vf_mlp_chain_gen.hpp), compiles it with hipcc on
 * first use into a shared object next to the library (visfly_amd/_jit.py) and registers it here.  vf_mlp_forward, vf_mlp_backward_data,
 * vf_ppo_update and their relatives then ask the loaded plugins after the built-in classes.  A plugin compiled against other struct
 * layouts is refused (VF_EINVAL).  Loading the same path twice is a no-op.
 * r06 (no new entry point; ABI 10): a class carries its activations (VF_ACTIVATION_*); the twin critic (heads 1 / 1 behind a pass-through
 * input, vf_mlp_desc.identity_mask) has generated classes too, served by vf_mlp_forward / vf_mlp_backward_data / vf_twin_q_update; and a
 * generated ACTOR class gets one more plugin per env kind / action type / integrator / motor lag that holds its instances of the two
 * persistent launches of a horizon: vf_bptt_rollout / vf_bptt_reverse then serve its layer table (16 agents per wave with the sub-step
 * tape: N <= 16 384 per launch), as vf_ppo_rollout does from the roll-out plugin. */
int vf_chain_plugin_load(const char* path);
int vf_chain_plugin_count(void);
const char* vf_chain_plugin_name(int32_t i);
int vf_chain_plugin_set_enabled(int on); /* 0: the loaded plugins are not asked (A/B against the block-tile kernels); returns the previous setting */
int64_t vf_chain_plugin_launches(void); /* launches the plugins have served since the library was loaded (capability queries are not counted) */
int vf_mlp_pack_weights(const vf_mlp_desc* desc, const float* params, float* packed, vf_stream_t stream);
/* out1 == NULL: the caller does not need the value head -- the register-chained kernel then skips the value trunk
 * (its saved activations are left untouched); layer tables that run on the LDS kernel return VF_EUNSUPPORTED. */
int vf_mlp_forward(const vf_mlp_desc* desc, const float* params, const float* packed, const float* in0, const float* in1,
                   const float* in2, const float* in3, float* out0, float* out1, int32_t M, vf_stream_t stream);
/* The forward of n_steps consecutive blocks of M_step rows (inputs / outputs (n_steps M_step, w)) in ONE launch, every row computed
 * exactly as vf_mlp_forward computes it in a launch over its block alone (the rows-per-wave choice is made for M_step rows): the
 * per-step inference passes of a recorded horizon -- SHAC's target critics on (obs', a') of every step, shac.py:234-239 -- without H
 * latency-bound launches.  Inference only (no layer saves); M_step a multiple of 32; register-chained network classes only
 * (VF_EUNSUPPORTED otherwise: the caller loops vf_mlp_forward). */
int vf_mlp_forward_steps(const vf_mlp_desc* desc, const float* params, const float* packed, const float* in0, const float* in1,
                         const float* in2, float* out0, float* out1, int32_t M_step, int32_t n_steps, vf_stream_t stream);

/* Whole-network backward in ONE launch (+ one fold): what loss.backward() does for the actor-critic MLP
 * (PPO.py:286-287, BPTT.py:127-129).  Layers are listed in execution (reverse) order.  A 64-row tile
 * is owned by one block for ALL layers, so the data gradient a layer writes to dX is read back as dY
 * by later entries of the same block without any grid-wide synchronisation; each block keeps dW/db
 * of the current layer in MFMA accumulators across its tiles and writes ONE partial per layer into
 * its row of `partials`; a fixed-order fold then produces grad[0:n_fold] (deterministic).
 *   dY (M, ld_dy) upstream gradient, Y (M, ld_y) saved layer output for the ReLU mask or NULL,
 *   X (M, ld_x) saved layer input, dX (M, ld_dx) data gradient or NULL; need_dx: 0 none, 1 store,
 *   2 add into dX.  Column offsets are folded into the pointers.
 *   partials: vf_mlp_backward_partial_floats(desc, M) floats.  accumulate != 0: grad += fold.
 * For the reference-default network classes (see the register-chain images above) the same result is produced by
 * three launches instead: reverse chain in registers (data gradients, one wave per 32 rows), weight / bias gradients
 * with both MFMA operands read straight from the saved activations and masked gradients, table-driven fold; the
 * intermediate dX buffers then hold the ReLU-masked gradients. */
typedef struct vf_mlp_bwd_layer {
    int32_t K, No;
    int32_t need_dx;
    int32_t ld_dy, ld_y, ld_x, ld_dx;
    int32_t wb_off;              /* packed data-gradient weights of this layer (vf_mlp_layer.wb_off) */
    int32_t wq_off;              /* reverse-chain image of this layer (vf_mlp_layer.wq_off) */
    int32_t act;                 /* ABI 10 (was pad0): VF_ACTIVATION_* of the layer whose output Y is; 0 with Y != NULL = ReLU */
    int64_t w_off, b_off;        /* offsets into the flat parameter buffer == into a partial row */
    const float* dY;
    const float* Y;
    const float* X;
    float* dX;
} vf_mlp_bwd_layer;
typedef struct vf_mlp_bwd_desc {
    int32_t n_layers;
    int32_t n_fold;              /* number of leading parameters covered by the layers (row length of a partial) */
    vf_mlp_bwd_layer layer[VF_MLP_MAX_LAYERS];
} vf_mlp_bwd_desc;
int32_t vf_mlp_backward_blocks(int32_t M);
/* floats `partials` must hold for this layer table and M (covers both the block-tile kernel and, for the network
 * classes it is instantiated for, the register-chained reverse sweep + row-slab weight-gradient kernel) */
int64_t vf_mlp_backward_partial_floats(const vf_mlp_bwd_desc* desc, int32_t M);
int vf_mlp_backward(const vf_mlp_bwd_desc* desc, const float* packed, float* partials, float* grad, int32_t M,
                    int32_t accumulate, vf_stream_t stream);
/* The two halves of the register-chained backward, for callers that reduce the weight gradient over several batches
 * at once (BPTT: one weight-gradient launch per horizon instead of one per step, BPTT.py:107-129):
 *   vf_mlp_backward_data   reverse chain only (M <= 16 384 rows: 16 rows per wave, else 32 -- the two agree to rounding, ~1e-6 of
 *                          the output scale, not to the bit): leaves the ReLU-masked gradient of every hidden layer in its dY buffer and
 *                          (need_dx on the first layers) dLoss/d observation in dX; VF_EUNSUPPORTED if the layer table is
 *                          not an instantiated network class / variant (vf_mlp_backward_data_supported: 1 / 0)
 *   vf_mlp_weight_grad     dW / db of every listed layer from dY (masked, as left by vf_mlp_backward_data) and X over M
 *                          rows + fold into grad; any layer table.  With the buffers of n batches stored back to back
 *                          (row stride unchanged) one call with M = n * rows covers them all.
 *                          partials: vf_mlp_backward_partial_floats(desc, M) floats. */
int vf_mlp_backward_data_supported(const vf_mlp_bwd_desc* desc);
int vf_mlp_backward_data(const vf_mlp_bwd_desc* desc, const float* packed, int32_t M, vf_stream_t stream);
int vf_mlp_weight_grad(const vf_mlp_bwd_desc* desc, float* partials, float* grad, int32_t M, int32_t accumulate,
                       vf_stream_t stream);
/* ABI 10: vf_mlp_weight_grad for the layers whose bit is set in layer_mask (bit i = desc->layer[i]) only, on the row-slab plan of the WHOLE
 * table (partials sized as for the whole table): the gradient of a layer has the same bits whether it was formed by this call or by
 * vf_mlp_weight_grad -- two calls with complementary masks are the two buckets of a two-bucket gradient exchange (a data-parallel step
 * all-reduces the first bucket while the second is being formed). */
int vf_mlp_weight_grad_layers(const vf_mlp_bwd_desc* desc, float* partials, float* grad, int32_t M, int32_t accumulate,
                              uint32_t layer_mask, vf_stream_t stream);
/* same, and the fold also leaves sum(grad[i]^2) of the values it wrote as vf_mlp_weight_grad_fold_blocks(desc) fp64
 * partials in sumsq_partials (for vf_adam_cfg.sumsq_partials).  loss_stats (optional): one more block of the same fold
 * launch sums the loss-statistic partial rows a vf_ppo_update call with stats == NULL left in its scratch -- what
 * vf_ppo_loss / vf_ppo_update otherwise spend a launch of their own on (same reduction order, same results). */
typedef struct vf_stats_fold {
    const float* part;       /* n_rows x 16 partial rows (vf_ppo_update scratch) */
    int32_t n_rows, pad0;
    float* stats;            /* fp32[16] out, as vf_ppo_loss */
    float* d_log_std_out;    /* optional, as vf_ppo_loss_cfg */
    float* stats_accum;      /* optional, as vf_ppo_loss_cfg */
} vf_stats_fold;
int32_t vf_mlp_weight_grad_fold_blocks(const vf_mlp_bwd_desc* desc);
int vf_mlp_weight_grad_sumsq(const vf_mlp_bwd_desc* desc, float* partials, float* grad, int32_t M, int32_t accumulate,
                             double* sumsq_partials, const vf_stats_fold* loss_stats, vf_stream_t stream);

/* Row gather of the rollout buffer for one epoch's permutation (SB3 RolloutBuffer.get: indices = np.random.permutation,
 * utils/algorithms/common.py:161-215 mirror): dst_f[i, :] = src_f[perm[i], :] for up to 8 row-major fp32 fields of `rows`
 * rows in ONE launch, so that every minibatch of the epoch is a contiguous slice. */
#define VF_GATHER_MAX_FIELDS 8
typedef struct vf_gather_fields {
    int32_t n_fields;
    int32_t width[VF_GATHER_MAX_FIELDS];
    const float* src[VF_GATHER_MAX_FIELDS];
    float* dst[VF_GATHER_MAX_FIELDS];
} vf_gather_fields;
int vf_gather_rows(const vf_gather_fields* fields, const int64_t* perm, int64_t rows, vf_stream_t stream);

/* Post-step bookkeeping of the rollout loop (SB3 OnPolicyAlgorithm.collect_rollouts; utils/algorithms/PPO.py:146): the
 * TimeLimit bootstrap reward_out = reward + gamma * V(terminal_observation) where the episode ended by truncation
 * (done && ep_flags bit 1, vf_env_out.ep_flags), and the next step's episode_start flags = float(done).  One launch instead
 * of a chain of elementwise tensor ops. */
int vf_rollout_post(const float* reward, const uint8_t* done, const uint8_t* ep_flags, const float* terminal_value, float gamma,
                    float* reward_out, float* next_episode_start, int32_t N, vf_stream_t stream);

/* The same bookkeeping with the bootstrap deferred to the end of the rollout: reward_out = reward, next_episode_start =
 * float(done), and every truncated agent's flat reward index (flat_base + i, flat_base = t * N) and terminal observation
 * row(s) (obs0 (N,w0), optional obs1 (N,w1)) are appended to a compact list through the device cursor (entries beyond
 * `capacity` are counted but not stored: the caller checks the cursor).  After the rollout the caller evaluates the value
 * head ONCE over the collected rows and vf_bootstrap_scatter performs rewards_flat[idx[j]] += gamma * values[j] -- the same
 * arithmetic as vf_rollout_post, without a second policy forward per step.  episode_stat (optional, (N,4) fp32, 16-byte
 * aligned): per-agent accumulators {episodes finished, sum of their returns, sum of their lengths, successes} advanced where
 * done (the rollout log of PPO._dump_logs, PPO.py:392-414; replaces the per-step vf_episode_stats launch). */
int vf_rollout_post_collect(const float* reward, const uint8_t* done, const uint8_t* ep_flags, float* reward_out,
                            float* next_episode_start, const float* obs0, const float* obs1, int32_t w0, int32_t w1, int32_t* cursor,
                            int32_t capacity, int32_t* idx_list, float* rows0, float* rows1, int32_t flat_base, int32_t N,
                            const float* ep_return, const int32_t* ep_length, float* episode_stat, vf_stream_t stream);
int vf_bootstrap_scatter(const int32_t* idx_list, const float* values, int32_t count, float gamma, float* rewards_flat,
                         vf_stream_t stream);

/* First-order policy optimisation (utils/algorithms/BPTT.py:107-129; the reparameterised squashed-Gaussian actor of
 * utils/policies/td_policies.py) with the action head fused into the chain kernels (reference-default policy shapes only;
 * VF_EUNSUPPORTED otherwise -> vf_mlp_forward + vf_reparam_fwd, vf_reparam_bwd + vf_mlp_backward_data):
 *   vf_mlp_forward_act        policy trunk only; writes action = tanh(mean + exp(log_std) * eps) (M,4), nothing else leaves
 *                             the chip besides the saved activations (same arithmetic as vf_reparam_fwd); obs_copy0/1
 *                             (optional): the observation rows are also written there -- the contiguous per-step copy a
 *                             horizon-wide vf_mlp_weight_grad reads as the first layers' X
 *   vf_mlp_backward_data_act  head gradient formed in the kernel: d_mean = d_action * (1 - action^2) is written to the action
 *                             head's dY buffer (the weight-gradient kernel reads it), g_log_std (M,4) += d_mean *
 *                             exp(log_std) * eps (same arithmetic as vf_reparam_bwd), then the reverse chain of
 *                             vf_mlp_backward_data */
int vf_mlp_forward_act(const vf_mlp_desc* desc, const float* params, const float* packed, const float* in0, const float* in1,
                       const float* log_std, const float* eps, float* action, float* obs_copy0, float* obs_copy1, int32_t M,
                       vf_stream_t stream);
int vf_mlp_backward_data_act(const vf_mlp_bwd_desc* desc, const float* packed, const float* d_action, const float* action,
                             const float* log_std, const float* eps, float* g_log_std, int32_t M, vf_stream_t stream);

/* Training-log statistics of one env step (PPO._dump_logs, utils/algorithms/PPO.py:392-414): acc4 (fp64, device) +=
 * {episodes finished this step, sum of their returns, sum of their lengths, successes}, from the step outputs
 * done / ep_return / ep_length / ep_flags (vf_env_out).  No host synchronisation. */
int vf_episode_stats(const uint8_t* done, const float* ep_return, const int32_t* ep_length, const uint8_t* ep_flags,
                     double* acc4, int32_t N, vf_stream_t stream);

/* First-order policy optimisation glue (utils/algorithms/BPTT.py:107-134), one launch each instead of a chain of
 * elementwise autograd nodes:
 *   vf_reparam_fwd     a = tanh(mean + exp(log_std) * eps)           (N,4) rows, log_std[4] shared
 *   vf_reparam_bwd     d_mean = d_action * (1 - a^2);  g_log_std (N,4) += d_mean * exp(log_std) * eps
 *   vf_bptt_accumulate loss += -reward * disc;  d_reward = -disc * scale;  disc <- disc * gamma * ~done + done  (:123-124) */
int vf_reparam_fwd(const float* mean, const float* log_std, const float* eps, float* action, int32_t N, vf_stream_t stream);
int vf_reparam_bwd(const float* d_action, const float* action, const float* log_std, const float* eps, float* d_mean,
                   float* g_log_std, int32_t N, vf_stream_t stream);
int vf_bptt_accumulate(const float* reward, const uint8_t* done, float* disc, float* loss, float* d_reward, float gamma,
                       float scale, int32_t N, vf_stream_t stream);
/* the same bookkeeping fused with the state checkpoint of the NEXT step (tape_row <- slab, slab_floats floats, both 16-byte
 * aligned): the two things the BPTT forward pass does between env step t and env step t + 1, as one launch */
int vf_bptt_accumulate_checkpoint(const float* reward, const uint8_t* done, float* disc, float* loss, float* d_reward, float gamma,
                                  float scale, int32_t N, const float* slab, float* tape_row, int64_t slab_floats, vf_stream_t stream);

/* Squashed diagonal Gaussian head (SB3 SquashedDiagGaussianDistribution as used by
 * policies.py:114,177-181,195-226): a = tanh(mean + exp(log_std) * eps), eps ~ N(0,1) from
 * Philox4x32-10 keyed by (seed, row, step); log_prob as SB3 computes it.  deterministic != 0: a = tanh(mean). */
int vf_head_sample(const float* mean, const float* log_std, float* action, float* log_prob, int32_t M,
                   uint64_t seed, uint64_t step, int32_t deterministic, vf_stream_t stream);

typedef struct vf_ppo_loss_cfg {
    float clip_range, ent_coef, vf_coef;
    float inv_batch;        /* 1 / global minibatch rows (means are over the global minibatch) */
    float* d_log_std_out;   /* optional: the 4 log_std gradients are also written here (tail of the flat gradient) */
    float* stats_accum;     /* optional: stats[0..15] are also added to this running fp32[16] accumulator */
    /* value-function clipping (PPO.py:237-243): values_pred = old_value + clamp(value - old_value, +-clip_range_vf);
     * old_value (M,) = the values stored at rollout time for THIS call's rows.  clip_range_vf <= 0 or old_value NULL: off */
    const float* old_value;
    float clip_range_vf;
    int32_t pad0;
    /* ABI 9, vf_ppo_update only (vf_ppo_loss ignores them): row_index != NULL -- row m of the call is row row_index[m] of in0 / in1 /
     * action / old_log_prob / ret / old_value, i.e. those pointers are the rollout buffer itself and no shuffled copy of it is made
     * (SB3's RolloutBuffer.get indexes the buffer with a permutation slice: buffers.py [SB3 2.2.1] :get/_get_samples).  adv stays in
     * call order: it is normalised per minibatch (PPO.py:215-220).  obs_copy0 / obs_copy1 (required with row_index): (M, in_dim)
     * buffers that receive the call's observation rows in call order -- the X operand of the first layers' weight gradients, i.e. what
     * bwd's first-layer entries must point to. */
    const int64_t* row_index;
    float* obs_copy0;
    float* obs_copy1;
} vf_ppo_loss_cfg;

/* Clipped-surrogate PPO loss and its gradient w.r.t. the head outputs (PPO.py:210-263;
 * evaluate_actions policies.py:228-254).  Inputs per row: mean[4], value, action[4] (squashed),
 * old_log_prob, normalised advantage, return; log_std[4] shared.
 * Outputs: d_mean (M,4), d_value (M,), and stats[16] (fp32, sums over THIS call's rows; divide by the
 * global batch): 0 policy_loss, 1 value_loss, 2 entropy_loss, 3 approx_kl, 4 clip_fraction,
 * 5..8 d_log_std[4] (already scaled by inv_batch).  scratch >= 16*1024 floats. */
int vf_ppo_loss(const float* mean, const float* value, const float* log_std, const float* action,
                const float* old_log_prob, const float* adv, const float* ret, float* d_mean, float* d_value,
                float* stats, int32_t M, const vf_ppo_loss_cfg* cfg, float* scratch, vf_stream_t stream);

typedef struct vf_adam_cfg {
    float lr, beta1, beta2, eps, weight_decay;
    float max_grad_norm;    /* <= 0: no clipping */
    int32_t step;           /* 1-based step count AFTER this update */
    int32_t pad0;
    /* optional: refresh the packed MLP weights (vf_mlp_pack_weights layout) in the same launch.  pack_map holds four
     * int32 per parameter: the float offsets of its copies in `packed` (forward / data-gradient / register-chain /
     * reverse-chain image), -1 = none */
    const int32_t* pack_map;
    float* packed;
    /* optional: the squared gradient norm as per-block partial sums left by vf_mlp_weight_grad (sumsq_partials != NULL
     * there) instead of grad_sumsq: total = sum of the n_sumsq_partials partials + sum of grad[i]^2 for i >= sumsq_tail_from
     * (the parameters the weight-gradient fold does not cover: log_std).  Saves the separate vf_sumsq launch. */
    const double* sumsq_partials;
    int32_t n_sumsq_partials;
    int32_t sumsq_tail_from;
} vf_adam_cfg;

/* ABI 10.  The tail of an optimiser step inside the weight-gradient launch (utils/algorithms/PPO.py:284-292: loss.backward()'s weight
 * gradients, clip_grad_norm_, optimizer.step()): vf_mlp_weight_grad_sumsq + vf_adam_step as ONE launch instead of three (weight
 * gradients, fold, Adam).  The waves of the weight-gradient kernel meet at device counters once their partials are out, fold the
 * partials chip-wide, form the squared gradient norm and run clip + Adam (+ the packed-weight refresh) on the elements they hold --
 * the same reduction orders and the same arithmetic as the separate launches: identical bits.
 *   tail->param / exp_avg / exp_avg_sq  the flat buffers of n parameters (as vf_adam_step)
 *   tail->adam                          as vf_adam_step; sumsq_partials = vf_mlp_weight_grad_fold_blocks(desc) doubles of workspace the
 *                                       launch writes and reads (n_sumsq_partials is ignored); sumsq_tail_from = first parameter the
 *                                       layer table does not cover (log_std): those take their gradient from grad[] as the caller /
 *                                       loss_stats->d_log_std_out left it
 *   tail->sync                          VF_WGRAD_SYNC_WORDS uint32, zeroed ONCE by the caller and then reused launch after launch by
 *                                       launches of one stream (the kernel leaves the counters at zero)
 *   loss_stats                          as vf_mlp_weight_grad_sumsq (part rows 16-byte aligned)
 * Needs every wave of the launch resident at once: VF_EUNSUPPORTED (reason in vf_last_error) when the plan for (desc, M) exceeds
 * what this device holds of the kernel or for streaming row counts (>= 131 072 with narrow layers) --
 * the caller then issues vf_mlp_weight_grad_sumsq + vf_adam_step.  A wave that waits longer than VISFLY_AMD_FUSED_TAIL_TIMEOUT_MS
 * (default 2000) sets sync[VF_WGRAD_SYNC_ABORT] and every waiter leaves WITHOUT the update: a caller polls that word at its
 * next host synchronisation and treats non-zero as a hard error (the counters are then stale: zero sync[] before reuse). */
#define VF_WGRAD_SYNC_WORDS 5120
#define VF_WGRAD_SYNC_ABORT 16
typedef struct vf_wgrad_tail {
    float* param;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
    vf_adam_cfg adam;
    uint32_t* sync;
} vf_wgrad_tail;
int vf_mlp_weight_grad_adam(const vf_mlp_bwd_desc* desc, float* partials, float* grad, int32_t M, int32_t accumulate,
                            const vf_stats_fold* loss_stats, const vf_wgrad_tail* tail, vf_stream_t stream);

/* One PPO minibatch step up to the weight gradients (utils/algorithms/PPO.py:203-287: evaluate_actions, the clipped-surrogate
 * loss, loss.backward()), for the network classes of the register-chained kernels: forward +
 * vf_ppo_loss + reverse chain in ONE launch (a wave carries 32 rows through the network, evaluates their loss terms on the
 * head outputs in its registers and walks back, masking with its own still-live activations), then the loss-statistic
 * fold.  fwd: the forward layer table with every `save` pointer set; bwd: the backward table (both trunks, no observation
 * gradient) whose head entries' dY buffers receive d_mean (M,4) / d_value (M,).  Afterwards the dY buffers hold the masked
 * layer gradients: call vf_mlp_weight_grad(bwd, ...) for dW / db.  stats / cfg / scratch as in vf_ppo_loss (scratch >=
 * 16 * ceil(M / 32) floats: one row of partial loss statistics per 32-row tile; until r05 M was capped at 32 768); stats == NULL: the partial rows stay in scratch for vf_mlp_weight_grad_sumsq's
 * loss_stats.  VF_EUNSUPPORTED: not an instantiated class -> use vf_mlp_forward, vf_ppo_loss,
 * vf_mlp_backward. */
int vf_ppo_update(const vf_mlp_desc* fwd, const vf_mlp_bwd_desc* bwd, const float* params, const float* packed,
                  const float* in0, const float* in1, const float* log_std, const float* action, const float* old_log_prob,
                  const float* adv, const float* ret, float* stats, int32_t M, const vf_ppo_loss_cfg* cfg, float* scratch,
                  vf_stream_t stream);

/* clip_grad_norm_ + torch.optim.Adam (weight_decay as L2) over one flat fp32 parameter buffer
 * (PPO.py:285-292).  grad_sumsq: device fp32[1] = sum of squares of the (already all-reduced) gradient,
 * produced by vf_sumsq.  */
int vf_sumsq(const float* x, int64_t n, float* out1, float* scratch /* >= 1024 floats */, vf_stream_t stream);
int vf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                 const float* grad_sumsq, const vf_adam_cfg* cfg, vf_stream_t stream);

/* =====================================================================================
 * Multi-GPU: the ONE exchange step of the path.  Agents shard by rank with no data-path collective; a
 * data-parallel PPO / BPTT optimiser step (utils/algorithms/PPO.py:285-292, BPTT.py:127-134) sums the flat fp32 gradient
 * buffer over the ranks -- one ncclAllReduce (RCCL over xGMI) per step, enqueued on the caller's stream.
 * Bootstrap: rank 0 calls vf_comm_unique_id, the 128 bytes travel through torch.distributed (or any side channel), every
 * rank calls vf_comm_init (collective).  RCCL is resolved from `path` at run time -- pass the librccl.so the process
 * already maps (PyTorch ships its own copy).
 * ===================================================================================== */
#define VF_COMM_ID_BYTES 128
/* The forward half of a BPTT horizon as ONE persistent launch (BPTT.py:107-124 is a closed loop: policy(obs_t) -> action_t ->
 * env.step -> obs_{t+1}): a wave owns 16 agents for all H steps -- 16-rows-per-wave policy chain + action head, state
 * checkpoint for the adjoint, fused env step with the agent in registers, loss / discount recurrence -- and leaves exactly what
 * H rounds of vf_mlp_forward_act + vf_env_step + vf_bptt_accumulate_checkpoint leave (bit-identical):
 *   desc          the policy's layer table with the `save` pointers of slot 0; the slots of the horizon are stored back to back
 *                 ([H][N][w] per buffer), so row t N + i addresses slot t (as for vf_mlp_weight_grad over a horizon)
 *   obs_slots0/1  [H][N][w] observation copies of the slots; slot 0 holds the current observation, slot t + 1 is written by
 *                 step t (obs_slots1 = the constant "target" rows of NavigationEnv, filled by the caller, or NULL)
 *   eps, actions  [H][N][4]: reparameterisation noise in, actions out (the tape's action rows)
 *   out           episode outputs as for vf_env_step; out->reward = N floats of scratch; obs / done are ignored
 *   obs_final     (N,13) observation after the last step;  tape [H] rows of tape_stride floats;  tape_done [H][N]
 *   d_reward      [H][N] = -disc_t * scale;  loss / disc (N,) in/out as for vf_bptt_accumulate
 * VF_EUNSUPPORTED unless: actor network of the built-in register-chained classes (or of a generated class whose BPTT plugin for this
 * configuration is loaded: vf_chain_plugin_load), Hover / Racing / Navigation env (r05: the *2 observation variants; r06: RacingEnv2's
 * 16 gate-relative columns), thrust / bodyrate actions, Euler or (repaired, utils/maths.py:353-386) RK4, with / without ctrl_delay, constant wind;
 * per-agent drag randomisation (dynamics.py:244-267) is carried in the slab's drag granules.
 *   substep_tape  optional (NULL: off), [H][S + 3][W][64] float4 with S = interval_steps and W = ceil(N / 16) waves: for every
 *                 (step, wave of 16 agents) S + 3 rows of 1 KiB.  Rows 0 .. S are component-major: the float4 at [k * 16 + m] of
 *                 a row holds component k (of w x y z; vectors as pure quaternions (0, x, y, z); rotor k) of four quantities of
 *                 agent slot m -- rows 0 .. S-1 the state at the head of each integrator sub-step: (q_k, v_k, w_k, rotor speed k);
 *                 row S the state after the last one before the clamps: (p_k, q_k, v_k, w_k) -- the layout both persistent
 *                 launches compute in (four lanes per agent).  Row S + 1, entry k of slot m at [k * 16 + m]: the step's other
 *                 inputs and its outcome: (body rates, ring-head bits) (angular acceleration, step-counter bits) (the action the
 *                 interval consumed) (done, d_reward, pre-step gate bits, 0); row S + 2, entries 0 / 1: the agent's drag
 *                 granules (drag randomisation only).  What autograd's tape keeps of dynamics.py:335-382; handed to vf_bptt_reverse, whose
 *                 adjoint then reads it (LDS-DMA, one step ahead) instead of replaying the S sub-steps (a third of its
 *                 instruction stream) and instead of the tape slab / done / d_reward rows (HBM-cold by then: two dependent
 *                 round trips at the head of every step).  16-byte aligned.
 * Actor classes: (a) the policy trunk of an actor-critic layer table with the state-independent `log_std` (4,) parameter
 * (policies.py:18-49; action = tanh(mean + exp(log_std) eps)); (b) the reference's own actor, utils/policies/td_policies.py:146-252
 * (what BPTT.py:113 / shac.py:219 call per step): a layer table with TWO 4-wide heads, latent_pi -> mu and log_latent_pi -> log_std,
 * action = tanh(mu + eps exp(clamp(log_std, VF_SAC_LOG_STD_MIN, VF_SAC_LOG_STD_MAX))) -- `log_std` is ignored (may be NULL) and
 *   log_std_rows  (H N, 4) out, required for (b): the second head of every step (vf_bptt_reverse's head reverse reads it)
 *   mean_rows     (H N, 4) out, optional (NULL: not kept): the first head of every step.  Both 16-byte aligned.
 * SHAC's horizon buffer (shac.py:259-266) additionally keeps, per step (both optional, NULL: not kept):
 *   reward_rows   [H][N] the reward of every step;  ep_flag_rows [H][N] bytes = out->ep_flags of the step, written where done
 *                 (needs out->ep_flags; shac.py:231-232 reads `episode_done` only there) */
int vf_bptt_rollout(vf_env* h, const vf_mlp_desc* desc, const float* params, const float* packed, const float* obs_slots0,
                    const float* obs_slots1, const float* log_std, const float* eps, float* actions, const vf_env_out* out,
                    float* obs_final, float* tape, int64_t tape_stride, uint8_t* tape_done, float* d_reward, float* loss,
                    float* disc, float gamma, float scale, int32_t H, float* substep_tape, float* mean_rows, float* log_std_rows,
                    float* reward_rows, uint8_t* ep_flag_rows, vf_stream_t stream);

/* ... and the reverse half (loss.backward() over the horizon, BPTT.py:127-129): for t = H-1 .. 0 the adjoint of env step t and the
 * policy's action-head reverse + reverse chain of step t, a wave owning 16 or 32 agents (the rows-per-wave choice
 * vf_mlp_backward_data makes for N rows) for the whole sweep; leaves what H rounds of
 * vf_env_step_bwd + vf_mlp_backward_data_act leave (bit-identical): the masked layer gradients of every slot (for ONE
 * vf_mlp_weight_grad over H N rows afterwards), d_mean rows, the per-row log_std gradient terms and the adjoint slab.
 *   desc       reverse layer table over the FLATTENED slots (H N rows; layer[0].dY = d_mean (H N, 4), the first-layer dX of the
 *              "state" branch = g_obs, (H N, 13) observation gradients);  d_action [H][N][4] scratch;  g_log_std [H][N][4] zeroed
 *   substep_tape  what vf_bptt_rollout wrote for the same H steps, or NULL (the interval is replayed; same results to the bit).
 *                 Read by the 16-agents-per-wave sweep only (N <= 16 384 per launch); ignored otherwise
 * Actor class (b) of vf_bptt_rollout: the table holds both trunks (layer[0] = the log_std head, dY = d_log_std (H N, 4) out; the
 * mean head's dY = d_mu (H N, 4) out), `log_std_rows` = what the forward launch saved; log_std / g_log_std are ignored (NULL);
 * what it leaves equals H rounds of vf_env_step_bwd + vf_shac_head_bwd + vf_mlp_backward_data.  16 agents per wave only
 * (N <= 16 384 per launch), VF_EUNSUPPORTED beyond.
 * Same restrictions as vf_bptt_rollout. */
int vf_bptt_reverse(vf_env* h, const vf_mlp_bwd_desc* desc, const float* packed, const float* log_std, const float* eps,
                    const float* actions, const float* tape, int64_t tape_stride, const uint8_t* tape_done, const float* d_reward,
                    float* adj_slab, float* d_action, const float* g_obs, float* g_log_std, int32_t H, const float* substep_tape,
                    const float* log_std_rows, vf_stream_t stream);

/* The same construction for PPO's collect_rollouts (SB3 OnPolicyAlgorithm.collect_rollouts, run by utils/algorithms/PPO.py:146;
 * n_steps rounds of policy.forward -> distribution.sample / log_prob -> env.step -> RolloutBuffer.add): T steps in ONE
 * persistent launch, a wave owning 16 or 32 agents (the rows-per-wave chain vf_mlp_forward runs N rows on, so that the heads are
 * the per-step path's to the bit).  Leaves what T rounds of vf_mlp_forward (both heads) + vf_head_sample (Philox counter
 * sample_step + 1 + t) + vf_env_step + vf_rollout_post_collect leave, bit for bit (tests/test_ppo_gpu.py): the rollout-buffer
 * rows, the compact TimeLimit-bootstrap list (rows in arbitrary order, as from the per-step calls), the per-agent episode
 * statistics, the episode outputs of the last step and the slab.  VF_EUNSUPPORTED unless: actor-critic network of the
 * register-chained classes, Hover / Navigation env with the raw-state observation, thrust / bodyrate actions, Euler or RK4
 * (BASELINE configs[2]'s dynamics incl. drag randomisation), ctrl_delay, constant wind.
 * desc: the policy's layer table WITHOUT activation copies (every layer's `save` NULL: inference only). */
typedef struct vf_ppo_rollout_args {
    int32_t T, w1, capacity, pad0;
    float* obs_state;           /* [T][N][13] RolloutBuffer "state" rows; row 0 = the current observation (caller), row t + 1 by step t */
    const float* obs_target;    /* [T][N][w1] "target" rows filled by the caller (NavigationEnv: constant), or NULL */
    const float* obs_target_row;/* (N,w1) the same for the bootstrap list, or NULL */
    float* obs_final;           /* (N,13) the observation after the last step */
    float* means;               /* [T][N][4] the head means (what the per-step vf_mlp_forward leaves), or NULL */
    float* values;              /* [T][N] */
    float* actions;             /* [T][N][4] */
    float* log_probs;           /* [T][N] */
    float* rewards;             /* [T][N] (before the bootstrap scatter) */
    float* episode_starts;      /* [T][N]: rows 1.. are written (row 0 = the caller's last episode starts) */
    float* last_starts;         /* (N,) episode starts after the last step */
    const float* log_std;       /* (4,) */
    uint64_t noise_key, sample_step;
    int32_t* cursor;            /* as for vf_rollout_post_collect */
    int32_t* idx_list;
    float* rows0;               /* [capacity][13] */
    float* rows1;               /* [capacity][w1] or NULL */
    float* stat;                /* (N,4) */
    const vf_env_out* out;      /* episode outputs; reward / done = N elements of scratch; obs ignored */
} vf_ppo_rollout_args;
int vf_ppo_rollout(vf_env* h, const vf_mlp_desc* desc, const float* params, const float* packed, const vf_ppo_rollout_args* a,
                   vf_stream_t stream);

/* ---- SHAC (utils/algorithms/shac.py:215-278; actor / twin critic of utils/policies/td_policies.py:82-252) -----------------
 * The networks are vf_mlp_desc layer tables like the PPO policy's (actor: two 4-wide heads mu / log_std over one extractor;
 * critic: features (+) action -> two Q trunks, the action columns entering through a frozen identity layer); the env step and
 * its adjoint are vf_env_step / vf_env_step_bwd; TD-lambda is vf_td_returns.  What is left of one iteration:
 *   vf_shac_head_fwd    action = tanh(mu + eps * exp(clamp(log_std, min, max)))   rows (N,4)   (td_policies.py:230-243 +
 *                       SB3 SquashedDiagGaussianDistribution.sample: Normal(mu, exp(log_std)).rsample(), tanh)
 *   vf_shac_head_bwd    its reverse: d_mu = d_action (1 - action^2), d_log_std = d_mu eps exp(log_std) inside the clamp
 *                       interval (closed, like torch.clamp), 0 outside
 *   vf_shac_accumulate  one horizon step of the actor objective INCLUDING the bootstrap term (shac.py:247-257):
 *                       next_value = min(q0, q1); loss -= reward disc; loss -= next_value disc gamma [(done | last_step) &
 *                       ~episode_done]; d_reward = -disc scale; disc = disc gamma ~done + done; writes the buffer rows
 *                       next_value_row (N,) and ep_done_row (N,) = done & (ep_flags & VF_EP_EPISODE_DONE) (shac.py:231-232)
 *   vf_twin_q_loss      loss_out[0] = sum((min(q0, q1) - target)^2) / M_global (mse_loss(target, values), :269) and its
 *                       gradient: dq_k = 2 (values - target) / M_global for the head that is the minimum (ties: q0), else 0;
 *                       scratch: vf_twin_q_loss_scratch_doubles(M) doubles.  M_global = rows summed over all ranks
 *   vf_polyak_update    target = target (1 - tau) + tau param   (SB3 polyak_update as shac.py:274 calls it) */
int vf_shac_head_fwd(const float* mu, const float* log_std, const float* eps, float* action, int32_t N, float log_std_min,
                     float log_std_max, vf_stream_t stream);
int vf_shac_head_bwd(const float* d_action, const float* action, const float* log_std, const float* eps, float* d_mu,
                     float* d_log_std, int32_t N, float log_std_min, float log_std_max, vf_stream_t stream);
int vf_shac_accumulate(const float* reward, const uint8_t* done, const uint8_t* ep_flags, const float* q0, const float* q1,
                       float* disc, float* loss, float* d_reward, float* next_value_row, uint8_t* ep_done_row, float gamma,
                       float scale, int32_t last_step, int32_t N, vf_stream_t stream);
/* vf_shac_accumulate for the H steps of a recorded horizon in one launch: rows (H, N) of reward / done / ep_flags / q0 / q1 in, d_reward /
 * next_value / ep_done rows out, disc / loss (N,) carried through the steps in order -- the same operations as H calls */
int vf_shac_accumulate_horizon(const float* reward, const uint8_t* done, const uint8_t* ep_flags, const float* q0, const float* q1,
                               float* disc, float* loss, float* d_reward, float* next_value, uint8_t* ep_done, float gamma, float scale,
                               int32_t H, int32_t N, vf_stream_t stream);
int64_t vf_twin_q_loss_scratch_doubles(int32_t M);
int vf_twin_q_loss(const float* q0, const float* q1, const float* target, float* dq0, float* dq1, float* loss_out,
                   double* scratch, int32_t M, int64_t M_global, vf_stream_t stream);
/* vf_twin_q_update     one critic update's per-row part in ONE launch (shac.py:267-270: values = critic(obs, action); mse_loss(returns,
 *                       min(Q1, Q2)); backward through both Q trunks and the critic's extractor): forward chain of the twin critic ->
 *                       vf_twin_q_loss's arithmetic on the head outputs still in registers -> reverse chain with the forward's own
 *                       activations as ReLU masks.  fwd / bwd: the critic's layer tables as for vf_mlp_forward (every layer input saved)
 *                       / vf_mlp_backward_data (no input gradient); in0 = the extractor's observation rows, in1 = the action rows;
 *                       leaves the saved inputs X, the masked gradients dZ and dQ1 / dQ2 (the head entries' dY) for
 *                       vf_mlp_weight_grad, loss_out[0] as vf_twin_q_loss.  scratch: vf_twin_q_update_scratch_doubles(M) doubles.
 *                       VF_EUNSUPPORTED when the tables are neither the built-in class (td_policies.ContinuousCritic over a
 *                       StateExtractor [128, 64], qf [64, 64]) nor a loaded generated twin-critic class (r06): the caller then runs
 *                       vf_mlp_forward / vf_twin_q_loss / vf_mlp_backward_data. */
int64_t vf_twin_q_update_scratch_doubles(int32_t M);
int vf_twin_q_update(const vf_mlp_desc* fwd, const vf_mlp_bwd_desc* bwd, const float* params, const float* packed, const float* in0,
                     const float* in1, const float* target, float* loss_out, double* scratch, int32_t M, int64_t M_global,
                     vf_stream_t stream);
int vf_polyak_update(float* target, const float* param, int64_t n, double tau, vf_stream_t stream);

/* test hook: fills the LDS of every CU with NaNs (tests/test_shac_gpu.py: layer tables with widths that are not multiples of
 * the 16-step MFMA chunk must not depend on what a previous kernel left in LDS) */
int vf_debug_poison_lds(vf_stream_t stream);

typedef struct vf_comm vf_comm;
int vf_comm_library(const char* path);
int vf_comm_unique_id(uint8_t* id /* VF_COMM_ID_BYTES, host */);
int vf_comm_init(const uint8_t* id, int32_t world, int32_t rank, vf_comm** out);
/* in-place sum over the ranks of buf[0..n): the flat gradient (+ loss statistics in its tail) / fp64 partial sums */
int vf_allreduce_grads(vf_comm* c, float* buf, int64_t n, vf_stream_t stream);
int vf_allreduce_f64(vf_comm* c, double* buf, int64_t n, vf_stream_t stream);
void vf_comm_destroy(vf_comm* c);

#ifdef __cplusplus
}
#endif
#endif
