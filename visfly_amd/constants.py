"""Host-side derivation of the kernel constants (``vf_dyn_cfg``, include/visfly_amd.h).

Mirrors, op for op and in fp32, what the reference computes once at construction time
(``Dynamics.__init__/_init/load/_get_scale_factor``, envs/base/dynamics.py:26-130,562-689)
so that the constants carry the same bits: e.g. the allocation-matrix inverse comes out
of the same LAPACK call (``B_inv[0,3] = 15.624999``, not 15.625).  The bits are checked
against the golden fixtures in tests/test_constants.py.
"""
from typing import Dict, Sequence, Union

import numpy as np
import torch as th

from .drone_params import load_params

ACTION_TYPES = {"thrust": 0, "bodyrate": 1, "velocity": 2, "position": 3}
INTEGRATORS = {"euler": 0, "rk4": 1}
GRAVITY = 9.81


def _cr_sqrt(x: th.Tensor) -> th.Tensor:
    """IEEE correctly rounded fp32 sqrt (torch's CPU fp32 sqrt is 1 ulp off for ~0.7 % of inputs)"""
    return th.sqrt(x.double()).float()


def _f(x) -> np.ndarray:
    return np.asarray(x.detach().cpu().numpy() if isinstance(x, th.Tensor) else x, dtype=np.float32)


def derive_constants(
        action_type: str = "bodyrate",
        dt: float = 0.005,
        ctrl_dt: float = 0.03,
        ctrl_delay: bool = True,
        comm_delay: float = 0.06,
        action_space: Sequence[float] = (-1, 1),
        integrator: str = "euler",
        cfg: Union[str, dict] = "drone_state",
        wind_settings=(0, 0, 0),
        transcendentals: str = "cr",
) -> Dict[str, np.ndarray]:
    """``transcendentals``: "cr" -- sin / cos / acos on the path are evaluated in fp64 and rounded once to fp32 (what the golden
    generator patches torch.sin / cos / acos to: the velocity / position controllers and NavigationEnv's view-angle term are
    then bit-identical to that reference); "sleef" -- SLEEF's u10 fp32 routines, the closest published algorithm to torch's
    closed-source MKL results (one ulp away for 2 - 8 % of the arguments), no fp64 arithmetic (include/visfly_amd.h VF_TRIG_*)"""
    if transcendentals not in ("cr", "sleef"):
        raise ValueError("transcendentals should be 'cr' or 'sleef'")
    if action_type not in ACTION_TYPES:
        raise AssertionError(f"action_type should be one of {list(ACTION_TYPES)}")
    if integrator not in INTEGRATORS:
        raise ValueError("type should be one of ['euler', 'rk4']")
    if not th.as_tensor(ctrl_dt) % th.as_tensor(dt) == 0:          # dynamics.py:71-72
        raise ValueError("ctrl_dt should be a multiple of dt")
    data = load_params(cfg)
    g = th.tensor([[0, 0, -GRAVITY]]).T                             # dynamics.py:15

    m = th.tensor(data["mass"])
    cross = th.tensor([data["cross_sections"]]).T
    k_quad = th.tensor([data["quad_drag_coeffs"]]).T * 0.5 * 1.225 * cross   # :567
    k_lin = th.tensor([data["linear_drag_coeffs"]]).T
    J = th.diag(th.tensor(data["inertia"]))                                   # :108
    J_inv = th.inverse(J)
    P = th.tensor(data["BODYRATE_PID"]["p"])
    Dm = th.tensor(data["BODYRATE_PID"]["d"])
    _ = data["THRUST_PID"], data["VELOCITY_PID"], data["POSITION_PID"]        # reference requires them (:574-576)
    kappa, arm = th.tensor(data["kappa"]), th.tensor(data["arm_length"])
    tm = th.tensor(data["thrust_map"])
    c_motor = th.exp(-th.tensor(1 / data["motor_tau"]) * dt)                  # :580-581
    w_max = data["motor_omega_max"]
    T_max = tm[0] * w_max ** 2 + tm[1] * w_max + tm[2]                        # :586-593
    rate_max, rate_min = th.tensor(data["max_rate"]), th.tensor(-data["max_rate"])
    acc_max, acc_min = (data["max_acc"] * -g[2]).clone(), th.tensor(0)        # :597-599

    # allocation matrix (:100-114)
    mdir = th.tensor([[1, -1, -1, 1.], [-1, -1, 1, 1], [0, 0, 0, 0.]])
    mdir = mdir / mdir.norm(dim=0)
    t_BM = arm * mdir
    B = th.vstack([th.ones(1, 4), t_BM[:2], kappa * th.tensor([1, -1, 1, -1])])
    B_inv = th.inverse(B)

    lo, hi = action_space
    acc_half = (acc_max - acc_min) / (hi - lo)                                # :627-631,650-654
    acc_mean = acc_max - acc_half * hi
    rate_half = (rate_max - rate_min) / (hi - lo)                             # :635-638
    rate_mean = rate_max - rate_half * hi
    # velocity / position action types (:660-689): the "velocity" Uniform holds the speed range in
    # velocity mode and the position range in position mode; the velocity-mode yaw Uniform is built
    # with half = yaw_bias (= 0, :671), i.e. the yaw channel is ignored there.
    rng_max = th.tensor(data["max_pos"] if action_type == "position" else data["max_spd"])
    vel_half = (rng_max - (-rng_max)) / (hi - lo)
    vel_mean = rng_max - vel_half * hi
    yaw_scale = th.as_tensor(th.pi - (-th.pi)) / (hi - lo)
    yaw_bias = th.pi - yaw_scale * hi
    yaw_half, yaw_mean = (yaw_scale, yaw_bias) if action_type == "position" else (yaw_bias, yaw_bias)
    geometric = action_type in ("velocity", "position")
    vel_pid, pos_pid = data["VELOCITY_PID"], data["POSITION_PID"]

    scale = 1 / (2 * tm[0])                                                   # :545

    def rotor_omega(T):                                                       # :545-553
        return scale * (-tm[1] + _cr_sqrt(tm[1].pow(2) - 4 * tm[0] * (tm[2] - T)))

    T_init = -(m * g / 4)[-1]                                                 # :85
    w_init = rotor_omega(T_init)                                              # :86

    if len(wind_settings) and isinstance(wind_settings[0], str):
        # string wind functions (:136-165): host lambdas of (t, previous wind), evaluated by Dynamics.update_wind() before every
        # step; the kernels then take per-agent rows (vf_dyn_set_wind) and this constant is not used
        wind = th.zeros((3, 1))
    else:
        wind = th.tensor(list(wind_settings)).reshape(3, 1) + th.zeros((3, 1))    # :135,172-174,388

    c = {
        "action_type": np.int32(ACTION_TYPES[action_type]),
        "integrator": np.int32(INTEGRATORS[integrator]),
        "interval_steps": np.int32(int(ctrl_dt / dt)),                        # :74
        "delay_steps": np.int32(int(comm_delay / ctrl_dt)),                   # :75
        "ctrl_delay": np.int32(bool(ctrl_delay)),
        "dt": np.float32(dt), "ctrl_dt": np.float32(ctrl_dt),
        "m": _f(m), "g_z": _f(g[2, 0]),
        "J": _f(J), "Jinv": _f(J_inv), "JP": _f(J @ P), "Dm": _f(Dm),
        "B": _f(B), "Binv": _f(B_inv),
        "c_motor": _f(c_motor), "one_minus_c": _f(1 - c_motor),
        "tm0": _f(tm[0]), "tm1": _f(tm[1]), "tm2": _f(tm[2]),
        "rot_scale": _f(scale), "rot_neg_tm1": _f(-tm[1]),
        "rot_tm1sq": _f(tm[1].pow(2)), "rot_4tm0": _f(4 * tm[0]),
        "T_min": np.float32(0), "T_max": _f(T_max),
        "acc_half": np.float32(0) if geometric else _f(acc_half[0]),
        "acc_mean": np.float32(0) if geometric else _f(acc_mean[0]),
        "rate_half": _f(rate_half) if action_type == "bodyrate" else np.float32(0),
        "rate_mean": _f(rate_mean) if action_type == "bodyrate" else np.float32(0),
        "k_lin": _f(k_lin[:, 0]), "k_quad": _f(k_quad[:, 0]),
        "wind": _f(wind[:, 0]),
        "pos_xy_lim": np.float32(100), "pos_z_lo": np.float32(0), "pos_z_hi": np.float32(20),   # :374-382
        "vel_lim": np.float32(20), "omg_lim": np.float32(10),
        "T_init": _f(T_init[0]), "w_init": _f(w_init[0]),
        "vel_half": _f(vel_half) if geometric else np.float32(0),
        "vel_mean": _f(vel_mean) if geometric else np.float32(0),
        "yaw_half": _f(yaw_half) if geometric else np.float32(0),
        "yaw_mean": _f(yaw_mean) if geometric else np.float32(0),
        "vel_p": _f(th.tensor(vel_pid["p"])), "vel_d": _f(th.tensor(vel_pid["d"])),
        "pos_d": _f(th.tensor(pos_pid["d"])),
        "Pm": _f(P), "P12": _f(1.2 * P),                                       # :451,491
        "trig_mode": np.int32(1 if transcendentals == "cr" else 0),
    }
    return {k: np.asarray(v) for k, v in c.items()}
