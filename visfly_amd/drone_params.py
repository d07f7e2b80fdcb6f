"""Physical / controller parameter sets of the supported airframes.

Same values as the reference's ``configs/drone/<name>.json`` files (consumed by
``Dynamics.load``, envs/base/dynamics.py:562-608), kept as one table of a shared base
plus per-airframe overrides.  ``load_params`` also accepts a path to a JSON file in the
reference's schema (including its ``BODYRAYE_PID`` spelling) so user configs drop in.
"""
import copy
import json
import os

_DIAG = lambda a, b, c: [[a, 0.0, 0.0], [0.0, b, 0.0], [0.0, 0.0, c]]
_ZERO3 = _DIAG(0.0, 0.0, 0.0)

_BASE = dict(
    mass=0.46,
    inertia=[0.00101, 0.00153, 0.00203],
    quad_drag_coeffs=[0.5, 0.5, 0.5],
    linear_drag_coeffs=[0.005, 0.005, 0.00575],
    cross_sections=[0.01, 0.01, 0.03],
    max_spd=10.0, max_pos=10.0,
    arm_length=0.075, kappa=0.016,
    motor_omega_max=3500, motor_omega_min=200, motor_tau=0.033,
    thrust_map=[4.0426e-07, 2.5583e-05, -0.026215],
    POSITION_PID=dict(p=1, i=0.01, d=0.25),
)

_SLOW_PID = dict(p=_DIAG(35.6, 38.6, 22.0), i=_ZERO3, d=_DIAG(0.0003, 0.0002, 0.0003))
_STATE_PID = dict(p=_DIAG(60.0, 50.6, 55.0), i=_ZERO3, d=_DIAG(0.001, 0.001, 0.002))
_ORIN_PID = dict(p=_DIAG(40.0, 47.6, 30.0), i=_ZERO3, d=_DIAG(0.00025, 0.0003, 0.0001))
_THRUST_PID = dict(p=1.0, i=0.0, d=0.0)
_VEL_FAST = dict(p=3.0, i=0.01, d=0.7)
_VEL_SLOW = dict(p=0.8, i=0.01, d=0.7)
_ORIN = dict(mass=0.671, inertia=[0.001373, 0.001834, 0.000867], max_rate=3.0, BODYRATE_PID=_ORIN_PID,
             THRUST_PID=_THRUST_PID, VELOCITY_PID=_VEL_FAST)

# airframes without max_acc / THRUST_PID raise KeyError exactly like the reference's load()
_VARIANTS = {
    "drone_state": dict(max_rate=2.0, max_acc=3.0, BODYRATE_PID=_STATE_PID, THRUST_PID=_THRUST_PID,
                        VELOCITY_PID=_VEL_FAST),
    "drone_state_fast": dict(max_rate=3.0, max_acc=2.0, BODYRATE_PID=_STATE_PID, VELOCITY_PID=_VEL_FAST),
    "drone_d435i": dict(max_rate=1.0, BODYRATE_PID=_SLOW_PID, VELOCITY_PID=_VEL_SLOW),
    "drone_d435i_n100": dict(mass=0.77, max_rate=1.0, BODYRATE_PID=_SLOW_PID, VELOCITY_PID=_VEL_SLOW),
    "drone_d435i_jetson_orin_nx": dict(_ORIN, max_acc=2.0),
    "drone_d435i_jetson_orin_nx_fast": dict(_ORIN, max_acc=3.0),
    "example": dict(inertia=[0.0002178, 0.0003299, 0.0004377], max_rate=1.0,
                    BODYRATE_PID=dict(p=_DIAG(45.0, 40.6, 40.0), i=_ZERO3, d=_DIAG(0.00017, 0.0002, 0.0003)),
                    VELOCITY_PID=_VEL_SLOW),
}


def available():
    return sorted(_VARIANTS)


def load_params(cfg="drone_state"):
    """cfg: airframe name, or path to a JSON file in the reference schema -> parameter dict"""
    if isinstance(cfg, dict):
        data = copy.deepcopy(cfg)
    elif cfg in _VARIANTS:
        data = copy.deepcopy(_BASE)
        data.update(copy.deepcopy(_VARIANTS[cfg]))
        data["name"] = cfg
    else:
        path = cfg if cfg.endswith(".json") else cfg + ".json"
        if not os.path.exists(path):
            raise FileNotFoundError(f"drone cfg '{cfg}': not a built-in airframe {available()} and no such file")
        with open(path, "r") as f:
            data = json.load(f)
    if "BODYRAYE_PID" in data:  # the reference's JSON key is misspelled
        data["BODYRATE_PID"] = data.pop("BODYRAYE_PID")
    return data
