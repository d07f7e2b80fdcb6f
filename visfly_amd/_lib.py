"""ctypes binding of libvisfly_amd.so -- the C-ABI declared in include/visfly_amd.h.

The product path has NO fallback: if the HIP library is missing or cannot be loaded,
importing this module's ``lib()`` raises.
"""
import ctypes as C
import os

import numpy as np
import torch  # noqa: F401  -- must be imported first: it maps the one HIP runtime (libamdhip64.so.7) we share

from ._build import LIB

TILE = 64
# granule indices (include/visfly_amd.h): vectors live in components 1..3
G_POS, G_QUAT, G_VEL, G_OMG, G_MOT, G_THR, G_AACC, G_ACC, G_RING = range(9)


GEOMETRIC_FIELDS = ("vel_half", "vel_mean", "yaw_half", "yaw_mean", "vel_p", "vel_d", "pos_d", "Pm", "P12")


class DynCfg(C.Structure):
    """mirror of vf_dyn_cfg"""
    _fields_ = [
        ("action_type", C.c_int32), ("integrator", C.c_int32),
        ("interval_steps", C.c_int32), ("delay_steps", C.c_int32),
        ("ctrl_delay", C.c_int32), ("trig_mode", C.c_int32),
        ("dt", C.c_float), ("ctrl_dt", C.c_float),
        ("m", C.c_float), ("g_z", C.c_float),
        ("J", C.c_float * 9), ("Jinv", C.c_float * 9),
        ("JP", C.c_float * 9), ("Dm", C.c_float * 9),
        ("B", C.c_float * 16), ("Binv", C.c_float * 16),
        ("c_motor", C.c_float), ("one_minus_c", C.c_float),
        ("tm0", C.c_float), ("tm1", C.c_float), ("tm2", C.c_float),
        ("rot_scale", C.c_float), ("rot_neg_tm1", C.c_float),
        ("rot_tm1sq", C.c_float), ("rot_4tm0", C.c_float),
        ("T_min", C.c_float), ("T_max", C.c_float),
        ("acc_half", C.c_float), ("acc_mean", C.c_float),
        ("rate_half", C.c_float), ("rate_mean", C.c_float),
        ("k_lin", C.c_float * 3), ("k_quad", C.c_float * 3),
        ("wind", C.c_float * 3),
        ("pos_xy_lim", C.c_float), ("pos_z_lo", C.c_float), ("pos_z_hi", C.c_float),
        ("vel_lim", C.c_float), ("omg_lim", C.c_float),
        ("T_init", C.c_float), ("w_init", C.c_float),
        ("vel_half", C.c_float), ("vel_mean", C.c_float),
        ("yaw_half", C.c_float), ("yaw_mean", C.c_float),
        ("vel_p", C.c_float), ("vel_d", C.c_float), ("pos_d", C.c_float),
        ("Pm", C.c_float * 9), ("P12", C.c_float * 9),
    ]

    @classmethod
    def from_dict(cls, d):
        c = cls()
        for name, _ in cls._fields_:
            if name == "trig_mode":       # VF_TRIG_CR unless the constants say otherwise (fixtures older than the field: no trig calls)
                c.trig_mode = int(d.get("trig_mode", 1))
                continue
            if name not in d and name in GEOMETRIC_FIELDS:
                continue                     # constants of the velocity/position controller: zero when unused
            v = d[name]
            cur = getattr(c, name)
            if isinstance(cur, (int, float)):
                setattr(c, name, np.asarray(v).item())
            else:
                arr = np.asarray(v, dtype=np.float32).reshape(-1)
                if arr.size != len(cur):
                    raise ValueError(f"constant '{name}': expected {len(cur)} values, got {arr.size}")
                for i, x in enumerate(arr):
                    cur[i] = float(x)
        return c


EUNSUPPORTED = -4         # VF_EUNSUPPORTED
ABI_VERSION = 10         # VF_ABI_VERSION of include/visfly_amd.h this binding mirrors
MAX_GATES, MAX_SPAWN = 8, 4


class SpawnBox(C.Structure):
    """mirror of vf_spawn_box"""
    _fields_ = [(n, C.c_float * 3) for n in ("pos_mean", "pos_half", "ori_mean", "ori_half",
                                              "vel_mean", "vel_half", "omg_mean", "omg_half")]


class EnvCfg(C.Structure):
    """mirror of vf_env_cfg"""
    _fields_ = [
        ("kind", C.c_int32), ("max_episode_steps", C.c_int32),
        ("is_collision_reset", C.c_int32), ("n_gates", C.c_int32),
        ("bbox_lo", C.c_float * 3), ("bbox_hi", C.c_float * 3),
        ("uav_radius", C.c_float), ("success_radius", C.c_float),
        ("target", C.c_float * 3), ("gates", (C.c_float * 3) * MAX_GATES),
        ("n_spawn", C.c_int32), ("drag_random", C.c_float),
        ("spawn", SpawnBox * MAX_SPAWN),
        ("seed", C.c_uint64),
        ("obs_mode", C.c_int32), ("reward_mode", C.c_int32),
        ("spawn_prefetch", C.c_int32), ("sense_radius", C.c_float),
    ]


class EnvOut(C.Structure):
    """mirror of vf_env_out (device pointers)"""
    _fields_ = [(n, C.c_void_p) for n in ("obs", "reward", "done", "ep_return", "ep_length", "ep_flags",
                                          "terminal_obs", "gate", "ep_past_gates", "terminal_gate", "done_list", "done_count")]


class PpoRolloutArgs(C.Structure):
    """mirror of vf_ppo_rollout_args"""
    _fields_ = ([("T", C.c_int32), ("w1", C.c_int32), ("capacity", C.c_int32), ("pad0", C.c_int32)] +
                [(n, C.c_void_p) for n in ("obs_state", "obs_target", "obs_target_row", "obs_final", "means", "values", "actions",
                                           "log_probs", "rewards", "episode_starts", "last_starts", "log_std")] +
                [("noise_key", C.c_uint64), ("sample_step", C.c_uint64)] +
                [(n, C.c_void_p) for n in ("cursor", "idx_list", "rows0", "rows1", "stat", "out")])


class EnvRollout(C.Structure):
    """mirror of vf_env_rollout"""
    _fields_ = [("actions", C.c_void_p), ("action_stride", C.c_int64), ("out", EnvOut),
                ("obs_stride", C.c_int64), ("reward_stride", C.c_int64), ("done_stride", C.c_int64),
                ("K", C.c_int32), ("auto_reset", C.c_int32)]


class EnvView(C.Structure):
    """mirror of vf_env_view (device pointers)"""
    _fields_ = [(n, C.c_void_p) for n in ("step_count", "rewards", "flags", "col_point", "col_vec", "col_dis",
                                          "gate", "past_gates")]


class EnvBwdArgs(C.Structure):
    """mirror of vf_env_bwd_args (device pointers)"""
    _fields_ = [(n, C.c_void_p) for n in ("tape_slab", "action", "d_obs", "d_reward", "done", "adj_slab", "d_action")]


MLP_MAX_LAYERS, MLP_MAX_BUFS, MLP_OUT0, MLP_OUT1 = 16, 16, 100, 101


class MlpLayer(C.Structure):
    """mirror of vf_mlp_layer"""
    _fields_ = [("K", C.c_int32), ("No", C.c_int32), ("relu", C.c_int32), ("src", C.c_int32), ("src_col", C.c_int32),
                ("dst", C.c_int32), ("dst_col", C.c_int32), ("w_off", C.c_int32), ("b_off", C.c_int32),
                ("save_ld", C.c_int32), ("wt_off", C.c_int32), ("wb_off", C.c_int32), ("wr_off", C.c_int32),
                ("wq_off", C.c_int32), ("save", C.c_void_p)]


class MlpDesc(C.Structure):
    """mirror of vf_mlp_desc"""
    _fields_ = [("n_layers", C.c_int32), ("n_inputs", C.c_int32), ("in_dim", C.c_int32 * 4),
                ("lds_off", C.c_int32 * MLP_MAX_BUFS), ("lds_stride", C.c_int32 * MLP_MAX_BUFS),
                ("lds_floats", C.c_int32), ("identity_mask", C.c_int32), ("layer", MlpLayer * MLP_MAX_LAYERS)]


class StatsFold(C.Structure):
    """mirror of vf_stats_fold"""
    _fields_ = [("part", C.c_void_p), ("n_rows", C.c_int32), ("pad0", C.c_int32), ("stats", C.c_void_p),
                ("d_log_std_out", C.c_void_p), ("stats_accum", C.c_void_p)]


class GatherFields(C.Structure):
    """mirror of vf_gather_fields"""
    _fields_ = [("n_fields", C.c_int32), ("width", C.c_int32 * 8), ("src", C.c_void_p * 8), ("dst", C.c_void_p * 8)]


class MlpBwdLayer(C.Structure):
    """mirror of vf_mlp_bwd_layer"""
    _fields_ = [("K", C.c_int32), ("No", C.c_int32), ("need_dx", C.c_int32), ("ld_dy", C.c_int32), ("ld_y", C.c_int32),
                ("ld_x", C.c_int32), ("ld_dx", C.c_int32), ("wb_off", C.c_int32), ("wq_off", C.c_int32), ("act", C.c_int32),
                ("w_off", C.c_int64), ("b_off", C.c_int64),
                ("dY", C.c_void_p), ("Y", C.c_void_p), ("X", C.c_void_p), ("dX", C.c_void_p)]


class MlpBwdDesc(C.Structure):
    """mirror of vf_mlp_bwd_desc"""
    _fields_ = [("n_layers", C.c_int32), ("n_fold", C.c_int32), ("layer", MlpBwdLayer * MLP_MAX_LAYERS)]


class PpoLossCfg(C.Structure):
    """mirror of vf_ppo_loss_cfg"""
    _fields_ = [("clip_range", C.c_float), ("ent_coef", C.c_float), ("vf_coef", C.c_float), ("inv_batch", C.c_float),
                ("d_log_std_out", C.c_void_p), ("stats_accum", C.c_void_p),
                ("old_value", C.c_void_p), ("clip_range_vf", C.c_float), ("pad0", C.c_int32),
                ("row_index", C.c_void_p), ("obs_copy0", C.c_void_p), ("obs_copy1", C.c_void_p)]


class AdamCfg(C.Structure):
    """mirror of vf_adam_cfg"""
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("max_grad_norm", C.c_float), ("step", C.c_int32), ("pad0", C.c_int32),
                ("pack_map", C.c_void_p), ("packed", C.c_void_p), ("sumsq_partials", C.c_void_p),
                ("n_sumsq_partials", C.c_int32), ("sumsq_tail_from", C.c_int32)]


class WgradTail(C.Structure):
    """mirror of vf_wgrad_tail"""
    _fields_ = [("param", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_int64),
                ("adam", AdamCfg), ("sync", C.c_void_p)]


WGRAD_SYNC_WORDS, WGRAD_SYNC_ABORT = 5120, 16


class VisflyError(RuntimeError):
    pass


_lib = None
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/visfly_amd.h declares
SIGNATURES = {
    "vf_last_error": (C.c_char_p, []),
    "vf_abi_version": (C.c_int32, []),
    "vf_dyn_create": (C.c_int, [C.POINTER(DynCfg), C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "vf_dyn_destroy": (None, [_vp]),
    "vf_dyn_granules": (C.c_int32, [_vp]),
    "vf_dyn_slab_floats": (C.c_int64, [_vp]),
    "vf_dyn_bind": (C.c_int, [_vp, _vp]),
    "vf_dyn_step": (C.c_int, [_vp, _vp, _vp, _vp]),
    "vf_dyn_reset": (C.c_int, [_vp, _vp, C.c_int32] + [_vp] * 10 + [_vp]),
    "vf_dyn_time_steps": (C.c_int, [_vp, _vp, _vp, C.c_int32, _vp, C.POINTER(C.c_float)]),
    "vf_env_create": (C.c_int, [C.POINTER(DynCfg), C.POINTER(EnvCfg), C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "vf_env_destroy": (None, [_vp]),
    "vf_env_granules": (C.c_int32, [_vp]),
    "vf_env_slab_floats": (C.c_int64, [_vp]),
    "vf_env_bind": (C.c_int, [_vp, _vp]),
    "vf_env_dyn": (_vp, [_vp]),
    "vf_env_reset": (C.c_int, [_vp, _vp, C.c_int32, _vp, _vp]),
    "vf_env_step": (C.c_int, [_vp, _vp, C.POINTER(EnvOut), C.c_int32, _vp]),
    "vf_env_step_n": (C.c_int, [_vp, C.POINTER(EnvRollout), _vp]),
    "vf_shac_head_fwd": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, C.c_float, C.c_float, _vp]),
    "vf_shac_head_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int32, C.c_float, C.c_float, _vp]),
    "vf_shac_accumulate": (C.c_int, [_vp] * 10 + [C.c_float, C.c_float, C.c_int32, C.c_int32, _vp]),
    "vf_twin_q_loss_scratch_doubles": (C.c_int64, [C.c_int32]),
    "vf_twin_q_loss": (C.c_int, [_vp] * 7 + [C.c_int32, C.c_int64, _vp]),
    "vf_shac_accumulate_horizon": (C.c_int, [_vp] * 10 + [C.c_float, C.c_float, C.c_int32, C.c_int32, _vp]),
    "vf_mlp_forward_steps": (C.c_int, [C.POINTER(MlpDesc)] + [_vp] * 7 + [C.c_int32, C.c_int32, _vp]),
    "vf_twin_q_update_scratch_doubles": (C.c_int64, [C.c_int32]),
    "vf_twin_q_update": (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpBwdDesc)] + [_vp] * 7 + [C.c_int32, C.c_int64, _vp]),
    "vf_polyak_update": (C.c_int, [_vp, _vp, C.c_int64, C.c_double, _vp]),
    "vf_debug_poison_lds": (C.c_int, [_vp]),
    "vf_bptt_reverse": (C.c_int, [_vp, C.POINTER(MlpBwdDesc)] + [_vp] * 5 + [C.c_int64] + [_vp] * 6 + [C.c_int32, _vp, _vp, _vp]),
    "vf_bptt_rollout": (C.c_int, [_vp, C.POINTER(MlpDesc)] + [_vp] * 7 + [C.POINTER(EnvOut), _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp,
                                  C.c_float, C.c_float, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vf_ppo_rollout": (C.c_int, [_vp, C.POINTER(MlpDesc), _vp, _vp, C.POINTER(PpoRolloutArgs), _vp]),
    "vf_dyn_step_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vf_env_ring_phase": (C.c_int32, [_vp]),
    "vf_env_set_ring_phase": (C.c_int, [_vp, C.c_int32]),
    "vf_dyn_ring_phase": (C.c_int32, [_vp]),
    "vf_dyn_set_ring_phase": (C.c_int, [_vp, C.c_int32]),
    "vf_env_rollout_fused": (C.c_int, [_vp, C.POINTER(EnvRollout), _vp]),
    "vf_env_graph_create": (C.c_int, [_vp, C.POINTER(EnvRollout), C.POINTER(_vp)]),
    "vf_env_graph_launch": (C.c_int, [_vp, _vp]),
    "vf_env_graph_destroy": (None, [_vp]),
    "vf_env_export_pose": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "vf_dyn_set_wind": (C.c_int, [_vp, _vp]),
    "vf_env_finish_step": (C.c_int, [_vp, _vp, _vp, C.POINTER(EnvOut), C.c_int32, _vp]),
    "vf_env_query": (C.c_int, [_vp, C.POINTER(EnvView), _vp]),
    "vf_env_time_steps": (C.c_int, [_vp, _vp, C.POINTER(EnvOut), C.c_int32, C.c_int32, _vp, C.POINTER(C.c_float)]),
    "vf_env_step_bwd": (C.c_int, [_vp, C.POINTER(EnvBwdArgs), _vp]),
    "vf_linear_bwd_weight_acc": (C.c_int, [_vp, C.c_int32, _vp, C.c_int32, _vp, C.c_int32, _vp, _vp, C.c_int32, C.c_int32,
                                           C.c_int32, _vp, C.c_int32, _vp]),
    "vf_gae": (C.c_int, [_vp] * 7 + [C.c_int32, C.c_int32, C.c_double, C.c_double, _vp]),
    "vf_td_returns": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_double, C.c_double, _vp]),
    "vf_adv_normalize": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, _vp, _vp, C.c_int32, _vp]),
    "vf_adv_normalize_segments": (C.c_int, [_vp, _vp, C.c_int32, C.c_int64, C.c_int64, _vp, C.c_int32, _vp]),
    "vf_linear_fwd": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp]),
    "vf_linear_bwd_data": (C.c_int, [_vp, C.c_int32, _vp, C.c_int32, _vp, _vp, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, _vp]),
    "vf_linear_bwd_scratch_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "vf_linear_bwd_weight": (C.c_int, [_vp, C.c_int32, _vp, C.c_int32, _vp, C.c_int32, _vp, _vp, C.c_int32, C.c_int32,
                                       C.c_int32, _vp, C.c_int32, _vp]),
    "vf_episode_stats": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int32, _vp]),
    "vf_reparam_fwd": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, _vp]),
    "vf_reparam_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int32, _vp]),
    "vf_bptt_accumulate": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float, C.c_int32, _vp]),
    "vf_bptt_accumulate_checkpoint": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float, C.c_int32, _vp, _vp, C.c_int64, _vp]),
    "vf_mlp_packed_floats": (C.c_int64, [C.POINTER(MlpDesc)]),
    "vf_mlp_pack_weights": (C.c_int, [C.POINTER(MlpDesc), _vp, _vp, _vp]),
    "vf_race_obs": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, _vp, C.c_int32, C.c_int32, C.c_float, _vp, _vp, C.c_int32, _vp]),
    "vf_chain_plugin_load": (C.c_int, [C.c_char_p]),
    "vf_chain_plugin_count": (C.c_int, []),
    "vf_chain_plugin_name": (C.c_char_p, [C.c_int32]),
    "vf_chain_plugin_launches": (C.c_int64, []),
    "vf_chain_plugin_set_enabled": (C.c_int, [C.c_int]),
    "vf_mlp_forward": (C.c_int, [C.POINTER(MlpDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int32, _vp]),
    "vf_mlp_backward_blocks": (C.c_int32, [C.c_int32]),
    "vf_mlp_backward_partial_floats": (C.c_int64, [C.POINTER(MlpBwdDesc), C.c_int32]),
    "vf_mlp_backward": (C.c_int, [C.POINTER(MlpBwdDesc), _vp, _vp, _vp, C.c_int32, C.c_int32, _vp]),
    "vf_mlp_backward_data_supported": (C.c_int, [C.POINTER(MlpBwdDesc)]),
    "vf_mlp_backward_data": (C.c_int, [C.POINTER(MlpBwdDesc), _vp, C.c_int32, _vp]),
    "vf_mlp_forward_act": (C.c_int, [C.POINTER(MlpDesc)] + [_vp] * 9 + [C.c_int32, _vp]),
    "vf_mlp_backward_data_act": (C.c_int, [C.POINTER(MlpBwdDesc)] + [_vp] * 6 + [C.c_int32, _vp]),
    "vf_mlp_weight_grad": (C.c_int, [C.POINTER(MlpBwdDesc), _vp, _vp, C.c_int32, C.c_int32, _vp]),
    "vf_mlp_weight_grad_layers": (C.c_int, [C.POINTER(MlpBwdDesc), _vp, _vp, C.c_int32, C.c_int32, C.c_uint32, _vp]),
    "vf_mlp_weight_grad_fold_blocks": (C.c_int32, [C.POINTER(MlpBwdDesc)]),
    "vf_mlp_weight_grad_sumsq": (C.c_int, [C.POINTER(MlpBwdDesc), _vp, _vp, C.c_int32, C.c_int32, _vp, C.POINTER(StatsFold), _vp]),
    "vf_mlp_weight_grad_adam": (C.c_int, [C.POINTER(MlpBwdDesc), _vp, _vp, C.c_int32, C.c_int32, C.POINTER(StatsFold), C.POINTER(WgradTail), _vp]),
    "vf_head_sample": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, C.c_uint64, C.c_uint64, C.c_int32, _vp]),
    "vf_ppo_loss": (C.c_int, [_vp] * 10 + [C.c_int32, C.POINTER(PpoLossCfg), _vp, _vp]),
    "vf_gather_rows": (C.c_int, [C.POINTER(GatherFields), _vp, C.c_int64, _vp]),
    "vf_rollout_post": (C.c_int, [_vp, _vp, _vp, _vp, C.c_float, _vp, _vp, C.c_int32, _vp]),
    "vf_rollout_post_collect": (C.c_int, [_vp] * 7 + [C.c_int32, C.c_int32, _vp, C.c_int32, _vp, _vp, _vp, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]),
    "vf_bootstrap_scatter": (C.c_int, [_vp, _vp, C.c_int32, C.c_float, _vp, _vp]),
    "vf_ppo_update": (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpBwdDesc)] + [_vp] * 10 + [C.c_int32, C.POINTER(PpoLossCfg), _vp, _vp]),
    "vf_sumsq": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp]),
    "vf_comm_library": (C.c_int, [C.c_char_p]),
    "vf_comm_unique_id": (C.c_int, [_vp]),
    "vf_comm_init": (C.c_int, [_vp, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "vf_allreduce_grads": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "vf_allreduce_f64": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "vf_comm_destroy": (None, [_vp]),
    "vf_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, _vp, C.POINTER(AdamCfg), _vp]),
}


def _single_hip_runtime():
    """two HIP runtimes in one process cannot share streams/pointers: refuse loudly"""
    seen = set()
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                seen.add(line.split()[-1])
    return seen


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise VisflyError(
                f"{LIB} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  visfly_amd has no CPU/eager fallback.")
        L = C.CDLL(LIB)
        rts = _single_hip_runtime()
        if len(rts) > 1:
            raise VisflyError(f"more than one HIP runtime mapped into this process: {sorted(rts)}")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.vf_abi_version() != ABI_VERSION:
            raise VisflyError("libvisfly_amd.so ABI version mismatch; rebuild")
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise VisflyError(f"libvisfly_amd error {rc}: {lib().vf_last_error().decode()}")


_warned = set()


def warn_unsupported(what: str):
    """a persistent launch answered VF_EUNSUPPORTED and the caller falls back to its launch-by-launch path (same results, roughly
    twice the time per step): say so ONCE per entry point, with the library's reason, instead of silently"""
    if what not in _warned:
        _warned.add(what)
        import warnings
        warnings.warn(f"visfly_amd: {what} is not available for this configuration ({lib().vf_last_error().decode()}); "
                      "falling back to one launch per step (same results, about half the rate)", stacklevel=3)


def ptr(t):
    """device pointer of a contiguous CUDA(ROCm) tensor, or NULL"""
    if t is None:
        return None
    if not t.is_cuda:
        raise VisflyError("visfly_amd kernels need device tensors (got a CPU tensor)")
    if not t.is_contiguous():
        raise VisflyError("visfly_amd kernels need contiguous tensors")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream(device):
    """raw hipStream_t of torch's current stream on `device` (no Python Stream object on the hot path)"""
    if _raw_stream is not None:
        idx = device.index if isinstance(device, torch.device) else torch.device(device).index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream
