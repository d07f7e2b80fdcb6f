#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- golden-vector generator.  Runs ONLY in the build container.

Imports the reference implementation read-only from /root/reference, runs it on CPU
and writes small input/output fixtures to tests/golden/*.npz.  Nothing from the
reference (source, bytecode) is copied: the fixtures are pure data (inputs, derived
constants as fp32 bit patterns, expected outputs).

"CR-sqrt oracle": this torch build's CPU fp32 ``sqrt`` is not IEEE correctly rounded
(1 ulp off for ~0.7 % of inputs), which no independent implementation
can reproduce.  The generator therefore patches ``torch.sqrt`` to the correctly rounded
result (fp64 sqrt rounded to fp32) before stepping the reference.  The un-patched
reference's step-256 state is stored next to it so the induced drift can be reported.

Usage:  python oracle/gen_golden.py [--only NAME]
"""
import argparse
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REF_PARENT = "/root"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
CHECKPOINTS = (1, 2, 4, 8, 16, 32, 64, 128, 256)

import torch as th  # noqa: E402

_orig_sqrt = th.sqrt


def cr_sqrt(x):
    return _orig_sqrt(x.double()).float() if x.dtype == th.float32 else _orig_sqrt(x)


_orig_unary = {n: (getattr(th, n), getattr(th.Tensor, n)) for n in ("sin", "cos", "acos")}


def _cr_unary(orig):
    """fp32 tensor -> the fp64 result rounded once to fp32 (differentiable: autograd sees double ops between two casts)"""
    def f(x, *a, **k):
        if isinstance(x, th.Tensor) and x.dtype == th.float32 and not a and not k:
            return orig(x.double()).float()
        return orig(x, *a, **k)
    return f


def use_cr_trig(on=True):
    """"CR-trig oracle" (r03, same precedent as CR-sqrt): this torch build's CPU fp32 sin / cos / acos are Intel MKL VML (closed
    source, not correctly rounded: 4.8 % / 5.0 % / 6.5 % of random arguments differ from the rounded fp64 result) and cannot be
    restated.  The generator patches the three -- module functions and Tensor methods, every call form the reference uses
    (envs/base/dynamics.py:432,438,467,473; utils/maths.py:256-276; envs/NavigationEnv.py:91) -- to fp64 evaluation rounded once
    to fp32, which the oracle and the HIP kernels reproduce bit for bit in their "cr" transcendental mode (oracle/vf_sleef.h).
    torch.atan2 stays torch's: it is SLEEF's atan2f_u10, restated bit for bit."""
    for n, (fn, meth) in _orig_unary.items():
        setattr(th, n, _cr_unary(fn) if on else fn)
        setattr(th.Tensor, n, _cr_unary(meth) if on else meth)


def use_cr_sqrt(on=True):
    """the CR oracle: sqrt AND sin / cos / acos (use_cr_trig) patched together; off = the reference exactly as torch runs it"""
    th.sqrt = cr_sqrt if on else _orig_sqrt
    use_cr_trig(on)


# ----------------------------------------------------------------------------------
# L0: Dynamics
# ----------------------------------------------------------------------------------

def import_dynamics():
    if REF_PARENT not in sys.path:
        sys.path.insert(0, REF_PARENT)
    from reference.envs.base.dynamics import Dynamics
    return Dynamics


def f32(x):
    return np.asarray(x.detach().cpu().numpy() if hasattr(x, "detach") else x, dtype=np.float32)


def extract_consts(d):
    """Derived constants of a reference Dynamics object -> dict of float32 arrays.

    Keys follow oracle/vf_oracle.h::vfo_consts; each is computed with the SAME torch ops
    the reference applies at run time so the bits are the reference's bits."""
    RD = sys.modules[type(d).__module__]   # reference dynamics module (holds the global g)
    tm = d._thrust_map
    at = d.action_type.name
    is_bodyrate = at == "BODYRATE"
    geometric = at in ("VELOCITY", "POSITION")
    npar = d._normal_params
    zero = th.zeros(1)
    c = {
        "action_type": np.int32({"THRUST": 0, "BODYRATE": 1, "VELOCITY": 2, "POSITION": 3}[at]),
        "integrator": np.int32(1 if d._integrator == "rk4" else 0),
        "interval_steps": np.int32(d._interval_steps),
        "delay_steps": np.int32(d._comm_delay_steps),
        "ctrl_delay": np.int32(bool(d._ctrl_delay)),
        "dt": np.float32(d.dt), "ctrl_dt": np.float32(d.ctrl_dt),
        "m": f32(d.m), "g_z": f32(RD.g[2, 0]),
        "J": f32(d._inertia), "Jinv": f32(d._inertia_inv),
        "JP": f32(d._inertia @ d._BODYRATE_PID.p), "Dm": f32(d._BODYRATE_PID.d),
        "B": f32(d._B_allocation), "Binv": f32(d._B_allocation_inv),
        "c_motor": f32(d._c), "one_minus_c": f32(1 - d._c),
        "tm0": f32(tm[0]), "tm1": f32(tm[1]), "tm2": f32(tm[2]),
        "rot_scale": f32(1 / (2 * tm[0])), "rot_neg_tm1": f32(-tm[1]),
        "rot_tm1sq": f32(tm[1].pow(2)), "rot_4tm0": f32(4 * tm[0]),
        "T_min": np.float32(d._bd_thrust.min), "T_max": f32(d._bd_thrust.max),
        "acc_half": np.float32(0) if geometric else f32(npar["acc"].half[0]),
        "acc_mean": np.float32(0) if geometric else f32(npar["acc"].mean[0]),
        "rate_half": f32(d._normal_params["bodyrate"].half[0]) if is_bodyrate else np.float32(0),
        "rate_mean": f32(d._normal_params["bodyrate"].mean[0]) if is_bodyrate else np.float32(0),
        "k_lin": f32(d._linear_drag_coeffs_mean[:, 0]), "k_quad": f32(d._quad_drag_coeffs_mean[:, 0]),
        "wind": f32(d.wind_velocity[:, 0]),
        "pos_xy_lim": np.float32(100), "pos_z_lo": np.float32(0), "pos_z_hi": np.float32(20),
        "vel_lim": np.float32(20), "omg_lim": np.float32(10),
        "T_init": f32(d._init_thrust[0]), "w_init": f32(d._init_motor_omega[0]),
        # geometric controller of the velocity / position action types (dynamics.py:414-496)
        "vel_half": f32(npar["velocity"].half.reshape(-1)[0]) if geometric else np.float32(0),
        "vel_mean": f32(npar["velocity"].mean.reshape(-1)[0]) if geometric else np.float32(0),
        "yaw_half": f32(npar["yaw"].half.reshape(-1)[0]) if geometric else np.float32(0),
        "yaw_mean": f32(npar["yaw"].mean.reshape(-1)[0]) if geometric else np.float32(0),
        "vel_p": f32(d._VELOCITY_PID.p), "vel_d": f32(d._VELOCITY_PID.d), "pos_d": f32(d._POSITION_PID.d),
        "Pm": f32(d._BODYRATE_PID.p), "P12": f32(1.2 * d._BODYRATE_PID.p),
        "trig_mode": np.int32(1),          # the fixtures are generated under use_cr_trig (VF_TRIG_CR)
    }
    return {k: np.asarray(v) for k, v in c.items()}


def decode_actions(q, hover, scale):
    """int8 fixture -> fp32 actions (platform-stable fp32 ops; same code in tests/_golden.py)"""
    a = q.astype(np.float32) * np.float32(scale / 127.0) + np.asarray(hover, np.float32)
    return np.clip(a, np.float32(-1), np.float32(1)).astype(np.float32)


def spawn_full_state(rng, N, consts):
    """random but platform-stable initial full_state (N,22): Hover spawn box, small tilt/vel."""
    fs = np.zeros((N, 22), np.float32)
    fs[:, 0:3] = (np.array([1, 0, 1.5]) + rng.uniform(-1, 1, (N, 3)) * np.array([1, 1, .5])).astype(np.float32)
    qv = rng.normal(size=(N, 4)) * np.array([0, .1, .1, .1]) + np.array([1, 0, 0, 0])
    qv /= np.linalg.norm(qv, axis=1, keepdims=True)
    fs[:, 3:7] = qv.astype(np.float32)
    fs[:, 7:10] = rng.uniform(-.5, .5, (N, 3)).astype(np.float32)
    fs[:, 10:13] = rng.uniform(-.3, .3, (N, 3)).astype(np.float32)
    fs[:, 13:17] = consts["w_init"]
    fs[:, 17:21] = consts["T_init"]
    fs[:, 21] = 0
    return fs


def extend_state(d):
    return f32(d.extend_state)


def run_dyn(Dynamics, kwargs, fs0, actions, checkpoints):
    N = fs0.shape[0]
    d = Dynamics(num=N, **kwargs)
    t = th.from_numpy
    d.reset(pos=t(fs0[:, 0:3].copy()), ori=t(fs0[:, 3:7].copy()), vel=t(fs0[:, 7:10].copy()),
            ori_vel=t(fs0[:, 10:13].copy()), motor_omega=t(fs0[:, 13:17].copy()),
            thrusts=t(fs0[:, 17:21].copy()), t=t(fs0[:, 21].copy()))
    out, obs = {}, {}
    for k in range(actions.shape[0]):
        s = d.step(t(actions[k].copy()))
        if (k + 1) in checkpoints:
            out[k + 1] = extend_state(d)
            obs[k + 1] = f32(s)
    return d, out, obs


def repair_rk4(module="reference.utils.maths"):
    """SURVEY App. C-1 minimal repair of Integrator.integrate(type='rk4') as a runtime patch
    of the imported reference (nothing written to /root/reference).  (i) wind passed to each
    stage, (ii) `d_* @ ks` restated as explicit elementwise weighted sums, (iii) the weighted
    d_ori_vel returned.  Fixtures produced with it are labelled repaired-oracle."""
    import importlib
    M = importlib.import_module(module)     # "VisFly.utils.maths" when the env layer was imported through the stubs
    Integrator = M.Integrator
    if getattr(Integrator, "_vf_repaired", False):
        return
    Integrator._vf_repaired = True
    orig = Integrator.integrate

    def integrate(pos, ori, vel, ori_vel, acc, tau, J, J_inv, dt, wind=th.zeros([3, 1]), type="euler"):
        if type != "rk4":
            return orig(pos, ori, vel, ori_vel, acc, tau, J, J_inv, dt, wind=wind, type=type)
        ks = th.tensor([1., 2., 2., 1.]) / 6
        sl = th.tensor([0.5, 0.5, 1])
        dp, dq, dv, dw = [], [], [], []
        oc, vc, wc = ori.clone(), vel.clone(), ori_vel.clone()
        for i in range(4):
            if i != 0:
                oc = ori + dq[i - 1] * sl[i - 1] * dt
                vc = vel + dv[i - 1] * sl[i - 1] * dt
                wc = ori_vel + dw[i - 1] * sl[i - 1] * dt
            a, b, c_, e = Integrator._get_derivatives(vel=vc, ori=oc, acc=acc, ori_vel=wc, tau=tau,
                                                      J=J, J_inv=J_inv, wind=wind)
            dp.append(a); dq.append(b); dv.append(c_); dw.append(e)
        ws = lambda k: ((k[0] * ks[0] + k[1] * ks[1]) + k[2] * ks[2]) + k[3] * ks[3]
        d_w = ws(dw)
        pos += ws(dp) * dt
        ori += ws(dq) * dt
        vel += ws(dv) * dt
        ori_vel += d_w * dt
        return pos, ori, vel, ori_vel, d_w

    Integrator.integrate = staticmethod(integrate)


DYN_CASES = {
    # name: (dynamics kwargs, hover action, noise scale)
    "dyn_bodyrate_euler": (dict(action_type="bodyrate", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True,
                                integrator="euler"), [-1 / 3, 0, 0, 0], 0.3),
    "dyn_bodyrate_euler_wide": (dict(action_type="bodyrate", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True,
                                     integrator="euler"), [-1 / 3, 0, 0, 0], 1.0),
    "dyn_thrust_euler": (dict(action_type="thrust", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True,
                              integrator="euler"), [-0.8333] * 4, 0.05),
    "dyn_bodyrate_nodelay": (dict(action_type="bodyrate", dt=0.0025, ctrl_dt=0.02, ctrl_delay=False,
                                  comm_delay=0.0, integrator="euler"), [-1 / 3, 0, 0, 0], 0.3),
    "dyn_bodyrate_dt005": (dict(action_type="bodyrate", dt=0.005, ctrl_dt=0.03, ctrl_delay=True,
                                integrator="euler", wind_settings=[0.5, -0.25, 0.125]), [-1 / 3, 0, 0, 0], 0.3),
    "dyn_bodyrate_rk4": (dict(action_type="bodyrate", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True,
                              integrator="rk4"), [-1 / 3, 0, 0, 0], 0.3),
    # string wind functions (dynamics.py:132-174,384-388): two triples of expressions in x = t and y = previous value, re-evaluated
    # at the top of every step.  Polynomials only: +, -, * round identically on the reference's CPU tensors and on GPU tensors
    "dyn_wind_functions": (dict(action_type="bodyrate", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True, integrator="euler",
                                wind_settings=["0.3 - 0.05*x", "0.02*x*x", "0.5*y + 0.1", "0*x + 0.125", "-0.01*x", "0.25*y - 0.05"]),
                           [-1 / 3, 0, 0, 0], 0.3),
    # geometric controller: sin/cos/atan2 are SLEEF in torch -> tolerance-level fixtures.
    # velocity: command = [yaw (ignored), v / 10 m/s]; position: [yaw / pi, p / 10 m]
    "dyn_velocity_euler": (dict(action_type="velocity", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True,
                                integrator="euler"), [0, 0.05, 0, 0], 0.15),
    "dyn_position_euler": (dict(action_type="position", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True,
                                integrator="euler"), [0.1, 0.1, 0, 0.15], 0.1),
}


def gen_dyn(name, N=128, steps=256, seed=1234):
    Dynamics = import_dynamics()
    kwargs, hover, scale = DYN_CASES[name]
    if kwargs.get("integrator") == "rk4":
        repair_rk4()
    rng = np.random.default_rng(seed)
    q = rng.integers(-127, 128, size=(steps, N, 4), dtype=np.int8)
    actions = decode_actions(q, hover, scale)
    use_cr_sqrt(True)
    d0 = Dynamics(num=N, **kwargs)
    consts = extract_consts(d0)
    fs0 = spawn_full_state(rng, N, consts)
    d, out, obs = run_dyn(Dynamics, kwargs, fs0, actions, CHECKPOINTS)
    use_cr_sqrt(False)
    _, raw, _ = run_dyn(Dynamics, kwargs, fs0, actions, (steps,))
    use_cr_sqrt(True)
    drift = float(np.abs(raw[steps][:, :13] - out[steps][:, :13]).max())
    print(f"{name}: N={N} steps={steps}; unpatched-vs-CR-sqrt drift @ {steps} = {drift:.3e}")
    save = {
        "actions_q": q, "hover": np.asarray(hover, np.float32), "scale": np.float32(scale),
        "fs0": fs0, "checkpoints": np.asarray(CHECKPOINTS, np.int32),
        "ext": np.stack([out[k] for k in CHECKPOINTS]),       # (9,N,28) extend_state
        "obs": np.stack([obs[k] for k in CHECKPOINTS]),       # (9,N,13) step() return
        "raw_ext_last": raw[steps],                            # un-patched torch.sqrt reference
        "label": np.asarray("repaired-oracle" if "rk4" in name else "cr-sqrt-oracle"),
    }
    # derived read-outs of the duck type at the last step: direction = body x axis (maths.py:123-133), Euler angles
    # (maths.py:244-249), accelerations and rotor state as the properties return them (dynamics.py:745-777)
    save.update(prop_direction=f32(d.direction), prop_euler=f32(d._orientation.toEuler().T), prop_acc=f32(d.acceleration),
                prop_ang_acc=f32(d.angular_acceleration), prop_motor_omega=f32(d.motor_omega), prop_thrusts=f32(d.thrusts),
                prop_t=f32(d.t), prop_velocity=f32(d.velocity))
    if isinstance(kwargs.get("wind_settings", [0])[0], str):
        save["wind_fn"] = np.asarray(kwargs["wind_settings"])
        save["wind_last"] = f32(d.wind_velocity)              # (3,N) after the last step
    save.update({"c_" + k: v for k, v in consts.items()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


# ----------------------------------------------------------------------------------
# L1-L3: env layer through stubs
# ----------------------------------------------------------------------------------

def _mod(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    m.__path__ = []
    sys.modules[name] = m
    return m


class _Auto(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        v = MagicMock()
        setattr(self, k, v)
        return v


def _auto(name):
    m = _Auto(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


_ENV_READY = False


def import_envs():
    """third-party stubs + fake SceneManager, then import the reference env classes"""
    global _ENV_READY
    pkgroot = "/tmp/vf_oracle_pkgroot"
    os.makedirs(pkgroot, exist_ok=True)
    link = os.path.join(pkgroot, "VisFly")
    if not os.path.islink(link):
        os.symlink("/root/reference", link)
    if pkgroot not in sys.path:
        sys.path.insert(0, pkgroot)
    if not _ENV_READY:
        class Box:
            def __init__(self, low, high, shape=None, dtype=np.float32):
                self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

        class Dict:
            def __init__(self, d):
                self.spaces = dict(d)

            def __getitem__(self, k):
                return self.spaces[k]

            def __setitem__(self, k, v):
                self.spaces[k] = v

            def keys(self):
                return self.spaces.keys()

        sp = _mod("gymnasium.spaces", Box=Box, Dict=Dict, Discrete=type("Discrete", (), {}), Space=object)
        for root in ("gymnasium", "gym"):
            _mod(root, spaces=sp, Space=object)
            _mod(root + ".spaces", **{k: v for k, v in sp.__dict__.items() if not k.startswith("__")})
            _mod(root + ".vector")
            _mod(root + ".vector.utils", spaces=sp)
        for n in ("stable_baselines3", "stable_baselines3.common", "stable_baselines3.common.vec_env"):
            _auto(n)
        sys.modules["stable_baselines3.common.vec_env"].VecEnv = type("VecEnv", (), {})
        ST = type("SensorType", (), dict(DEPTH="DEPTH", COLOR="COLOR", SEMANTIC="SEMANTIC"))
        _auto("habitat_sim").SensorType = ST
        _auto("habitat_sim.sensor").SensorType = ST
        for n in ("magnum", "quaternion", "cv2", "torchvision", "torchvision.models", "torchvision.transforms",
                  "torchvision.datasets", "imageio", "networkx", "torch.utils.tensorboard"):
            _auto(n)
        import VisFly.envs.base.droneEnv as DE

        class FakeSceneManager:  # visual=False never loads a scene
            def __init__(self, num_agent_per_scene=1, num_scene=1, col_refine_steps=0, **kw):
                self.num_scene, self.num_agent_per_scene = num_scene, num_agent_per_scene
                self.num_agent = num_scene * num_agent_per_scene
                self.col_refine_steps, self.scenes = col_refine_steps, [None] * num_scene
                self.sensor_settings = kw.get("sensor_settings", [])
                self.dynamic_object_position = self.dynamic_object_velocity = \
                    self.dynamic_object_acceleration = [[None] for _ in range(self.num_agent)]

            def close(self):
                pass

        DE.SceneManager = FakeSceneManager
        _ENV_READY = True
    from VisFly.envs.HoverEnv import HoverEnv
    from VisFly.envs.NavigationEnv import NavigationEnv
    from VisFly.envs.RacingEnv import RacingEnv

    class HoverEnvShim(HoverEnv):  # defect C-3: base passes predicted_obs
        def get_reward(self, predicted_obs=None):
            return super().get_reward()

    class RacingEnvShim(RacingEnv):  # defect C-4: signatures predate the base class, self.latent undefined
        def get_observation(self, indices=None, predicted_obs=None):
            from VisFly.utils.type import TensorDict
            return TensorDict({"state": self.state, "gate": self._next_target_i})

        def get_reward(self, predicted_obs=None):
            return super().get_reward()

        def reset(self, state=None, obs=None, **kw):
            return super().reset(state)

    return HoverEnvShim, NavigationEnv, RacingEnvShim


def import_racing2():
    """RacingEnv2 (envs/RacingEnv.py:218-267) under the same C-4 repairs as RacingEnv; its get_observation is the class's own"""
    import_envs()
    from VisFly.envs.RacingEnv import RacingEnv2

    class RacingEnv2Shim(RacingEnv2):
        def get_observation(self, indices=None, predicted_obs=None):
            return RacingEnv2.get_observation(self, indices)

        def get_reward(self, predicted_obs=None):
            return super().get_reward()

        def reset(self, state=None, obs=None, **kw):
            return super().reset(state)

    return RacingEnv2Shim


def import_envs2():
    """HoverEnv2 / NavigationEnv2: relative-position observations, Nav2 reward"""
    import_envs()
    from VisFly.envs.HoverEnv import HoverEnv2
    from VisFly.envs.NavigationEnv import NavigationEnv2

    class HoverEnv2Shim(HoverEnv2):  # same defect C-3 as HoverEnv: the base passes predicted_obs
        def get_reward(self, predicted_obs=None):
            from VisFly.envs.HoverEnv import HoverEnv
            return HoverEnv.get_reward(self)

    return HoverEnv2Shim, NavigationEnv2


ENV_DYN = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
RACING_DYN = dict(action_type="thrust", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
# gates moved next to the spawn boxes so that random flight passes them (the pass / advance / +20 logic)
RACING_TEST_GATES = [[2.1, 0.1, 1.0], [6., 2., 1.45], [6.1, -2., 1.5], [2., 2.1, 1.05]]

ENV_CASES = {
    # name: (env, ctor kwargs, hover, scale, steps)
    "env_hover": ("hover", dict(max_episode_steps=64), [-1 / 3, 0, 0, 0], 0.5, 256),
    "env_hover_256": ("hover", dict(max_episode_steps=256), [-1 / 3, 0, 0, 0], 0.3, 256),
    "env_nav": ("nav", dict(max_episode_steps=64, random_kwargs={"state_generator": {"class": "Uniform", "kwargs": [
        {"position": {"mean": [1., 0., 1.5], "half": [0., 2., 1.]}}]}}), [-0.2, 0, 0, 0], 0.6, 256),
    "env_racing": ("racing", dict(max_episode_steps=48), [-0.8333] * 4, 0.08, 256),
    "env_racing2": ("racing2", dict(max_episode_steps=48), [-0.8333] * 4, 0.08, 256),
    "env_nav_close": ("nav", dict(max_episode_steps=96, target=[2.5, 0., 1.5], random_kwargs={"state_generator": {
        "class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0.5, 1., 0.5]},
                                        "orientation": {"mean": [0., 0., 0.], "half": [0.2, 0.2, 1.0]},
                                        "velocity": {"mean": [1., 0., 0.], "half": [1., .5, .5]}}]}}),
                      [-0.3, 0, 0, 0], 0.5, 256),
    # observation / reward variants
    "env_hover2": ("hover2", dict(max_episode_steps=64, random_kwargs={"state_generator": {"class": "Uniform", "kwargs": [
        {"position": {"mean": [1., 0., 1.5], "half": [1.0, 1.0, 0.5]}}]}}), [-1 / 3, 0, 0, 0], 0.5, 256),
    "env_nav2": ("nav2", dict(max_episode_steps=96, target=[2.5, 0., 1.5], random_kwargs={"state_generator": {
        "class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0.5, 1., 0.5]},
                                        "orientation": {"mean": [0., 0., 0.], "half": [0.2, 0.2, 1.0]},
                                        "velocity": {"mean": [1., 0., 0.], "half": [1., .5, .5]}}]}}),
                 [-0.3, 0, 0, 0], 0.5, 256),
}


def gen_env(name, N=128, seed=42):
    HoverEnvShim, NavigationEnv, RacingEnv = import_envs()
    kind, kw, hover, scale, steps = ENV_CASES[name]
    use_cr_sqrt(True)
    cls = {"hover": HoverEnvShim, "nav": NavigationEnv, "racing": RacingEnv}.get(kind)
    if kind == "racing2":
        cls, kind = import_racing2(), "racing"        # everything but the observation is RacingEnv
    elif cls is None:
        H2, N2 = import_envs2()
        cls = {"hover2": H2, "nav2": N2}[kind]
    kw = dict(kw)
    if "target" in kw:
        kw["target"] = th.tensor(kw["target"])
    def make_env():
        e = cls(num_agent_per_scene=N, num_scene=1, seed=seed, visual=False,
                dynamics_kwargs=dict(RACING_DYN if kind == "racing" else ENV_DYN),
                device="cpu", **({"tensor_output": True} if kind in ("hover", "hover2", "nav2") else {}), **kw)
        e.tensor_output = True
        if kind == "racing":
            e.targets = th.as_tensor(RACING_TEST_GATES)
        return e

    rng = np.random.default_rng(seed + 1)
    q = rng.integers(-127, 128, size=(steps, N, 4), dtype=np.int8)
    actions = decode_actions(q, hover, scale)
    # the same run with the reference EXACTLY as torch runs it (MKL sin / cos / acos, this build's sqrt): reward / done of every step,
    # so that the tests can bound what the CR patches cost in faithfulness (the bit-level pins are against the patched
    # reference; this is the distance to the unpatched one)
    use_cr_sqrt(False)
    raw_env = make_env()
    raw_env.reset()
    raw_reward, raw_done = [], []
    for k in range(steps):
        _, r, d, _ = raw_env.step(th.from_numpy(actions[k].copy()))
        raw_reward.append(f32(r)); raw_done.append(d.numpy().astype(np.uint8))
    del raw_env
    use_cr_sqrt(True)
    env = make_env()
    consts = extract_consts(env.envs.dynamics)
    obs0 = env.reset()
    env_gate0 = env._next_target_i.clone().numpy().astype(np.int32) if kind == "racing" else None
    dyn = env.envs.dynamics
    rec = dict(reward=[], done=[], step_count=[], is_collision=[], is_out_bounds=[], success=[],
               col_dis=[], obs_state=[], ext_pre=[], gate=[], past=[])
    ev_step, ev_agent, ev_fs = [], [], []
    fs_init = f32(dyn.full_state)

    # capture pre-reset state: wrap examine
    pre = {}
    orig_examine = env.examine

    def examine():
        pre["ext"] = extend_state(dyn)
        pre["step_count"] = env._step_count.clone().numpy()
        pre["col_dis"] = f32(env.collision_dis)
        pre["is_collision"] = env.is_collision.clone().numpy()
        pre["is_out_bounds"] = env.is_out_bounds.clone().numpy()
        return orig_examine()

    env.examine = examine
    for k in range(steps):
        pre.clear()
        o, r, d, info = env.step(th.from_numpy(actions[k].copy()))
        didx = np.nonzero(d.numpy())[0]
        if len(didx):
            fs = f32(dyn.full_state)
            for i in didx:
                ev_step.append(k); ev_agent.append(i); ev_fs.append(fs[i])
                # what collect_info put into the info dict of the finished episode (droneGymEnv.py:238-275)
                ep = info[i]["episode"]
                rec.setdefault("ev_r", []).append(np.float32(ep["r"])); rec.setdefault("ev_l", []).append(np.int32(ep["l"]))
                rec.setdefault("ev_t", []).append(np.float32(ep["t"]))
                rec.setdefault("ev_tobs", []).append(f32(info[i]["terminal_observation"]["state"]))
                rec.setdefault("ev_flags", []).append(np.uint8(int(bool(info[i]["is_success"])) | (int(bool(info[i]["TimeLimit.truncated"])) << 1)
                                                               | (int(bool(info[i]["episode_done"])) << 2)
                                                               | (int(bool(ep["extra"]["collision"])) << 3)))
        rec["reward"].append(f32(r)); rec["done"].append(d.numpy().astype(np.uint8))
        if pre:
            rec["step_count"].append(pre["step_count"].astype(np.int32))
            rec["ext_pre"].append(pre["ext"]); rec["col_dis"].append(pre["col_dis"])
            rec["is_collision"].append(pre["is_collision"].astype(np.uint8))
            rec["is_out_bounds"].append(pre["is_out_bounds"].astype(np.uint8))
        else:
            rec["step_count"].append(env._step_count.clone().numpy().astype(np.int32))
            rec["ext_pre"].append(extend_state(dyn)); rec["col_dis"].append(f32(env.collision_dis))
            rec["is_collision"].append(env.is_collision.clone().numpy().astype(np.uint8))
            rec["is_out_bounds"].append(env.is_out_bounds.clone().numpy().astype(np.uint8))
        rec["success"].append(env._success.clone().numpy().astype(np.uint8))
        rec["obs_state"].append(f32(o["state"]))
        if "collision_vector" in o:      # NavigationEnv2's second observation entry, as returned
            rec.setdefault("obs_cv", []).append(f32(o["collision_vector"]))
        if "gate" in o:      # the gate index INSIDE the returned observation (not always env._next_target_i, see visfly_amd/envs/tasks.py)
            rec.setdefault("obs_gate", []).append(o["gate"].clone().numpy().astype(np.int32).reshape(N, -1))
            for i in didx:   # and inside the terminal observation of the agents that ended an episode in this step
                rec.setdefault("ev_tgate", []).append(int(np.asarray(info[i]["terminal_observation"]["gate"]).reshape(-1)[0]))
        if kind == "racing":   # post-auto-reset values, as returned in the observation
            rec["gate"].append(env._next_target_i.clone().numpy().astype(np.int32))
            rec["past"].append(env._past_targets_num.clone().numpy().astype(np.int32))
    print(f"{name}: N={N} steps={steps} resets={len(ev_step)} "
          f"(collisions {int(np.sum(rec['is_collision']))}, success {int(np.sum(rec['success']))})")
    # keep the fixture small: full pre-reset state only at sparse steps
    keep = sorted(set([0, 1, 2, 3, 7, 15, 31, 63, 64, 65, 127, 128, 191, 255, steps - 1]) & set(range(steps)))
    save = {
        "kind": np.asarray(kind), "max_episode_steps": np.int32(kw["max_episode_steps"]),
        "target": f32(env.target[0]) if kind != "racing" else np.zeros(3, np.float32), "seed": np.int32(seed),
        "actions_q": q, "hover": np.asarray(hover, np.float32), "scale": np.float32(scale),
        "fs_init": fs_init, "obs0_state": f32(obs0["state"]),
        "reward": np.stack(rec["reward"]), "done": np.stack(rec["done"]),
        "step_count": np.stack(rec["step_count"]), "is_collision": np.stack(rec["is_collision"]),
        "is_out_bounds": np.stack(rec["is_out_bounds"]), "success": np.stack(rec["success"]),
        "col_dis": np.stack(rec["col_dis"]),
        "keep_steps": np.asarray(keep, np.int32),
        "ext_pre_keep": np.stack([rec["ext_pre"][k] for k in keep]),
        "obs_state_keep": np.stack([rec["obs_state"][k] for k in keep]),
        "ev_step": np.asarray(ev_step, np.int32), "ev_agent": np.asarray(ev_agent, np.int32),
        "ev_fs": np.stack(ev_fs) if ev_fs else np.zeros((0, 22), np.float32),
        "spawn": np.asarray(repr(kw.get("random_kwargs", "hover-default"))),
        "label": np.asarray("repaired-oracle" if kind == "racing" else "cr-sqrt-oracle"),
        "raw_reward": np.stack(raw_reward), "raw_done": np.stack(raw_done),
    }
    if "ev_r" in rec:
        save.update(ev_r=np.asarray(rec["ev_r"], np.float32), ev_l=np.asarray(rec["ev_l"], np.int32), ev_t=np.asarray(rec["ev_t"], np.float32),
                    ev_tobs=np.stack(rec["ev_tobs"]), ev_flags=np.asarray(rec["ev_flags"], np.uint8))
    if "obs_cv" in rec:
        save["obs_cv"] = np.stack(rec["obs_cv"])
        save["obs0_cv"] = f32(obs0["collision_vector"])
    if "obs_gate" in rec:
        save["obs_gate"] = np.stack(rec["obs_gate"])
        save["obs0_gate"] = np.asarray(obs0["gate"]).astype(np.int32).reshape(N, -1)
        save["ev_tgate"] = np.asarray(rec.get("ev_tgate", []), np.int32)
    if kind == "racing":
        save.update(gates=np.asarray(RACING_TEST_GATES, np.float32), gate=np.stack(rec["gate"]), past=np.stack(rec["past"]),
                    gate0=env_gate0)
        print("   gate passes:", int(np.sum(np.diff(np.stack(rec["past"]), axis=0) > 0)))
    save.update({"c_" + k: v for k, v in consts.items()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


IMU_NOISE = {"IMU": {"model": "UniformNoiseModel", "kwargs": {
    "mean": [0.01, -0.02, 0.0, 0.0, 0.0, 0.0, 0.0, 0.05, 0.0, -0.05, 0.0, 0.01, 0.0],
    "half": [0.10, 0.10, 0.2, 0.05, 0.05, 0.05, 0.05, 0.30, 0.3, 0.30, 0.2, 0.20, 0.2]}}}


def gen_imu(name="env_hover_imu", N=128, steps=40, seed=42):
    """IMU noise with NON-ZERO amplitude (envs/base/droneEnv.py:99-125, utils/type.py:25-38): HoverEnv from seed 42, the
    noisy state sensor_obs["IMU"] after reset() and after every step (the draws come from the global generator between
    the spawn draws)"""
    HoverEnvShim, _, _ = import_envs()
    use_cr_sqrt(True)
    rk = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [1.0, 1.0, 0.5]}}]},
          "noise_kwargs": {k: {"model": v["model"], "kwargs": {a: th.tensor(b) for a, b in v["kwargs"].items()}}
                           for k, v in IMU_NOISE.items()}}
    env = HoverEnvShim(num_agent_per_scene=N, num_scene=1, seed=seed, visual=False, dynamics_kwargs=dict(ENV_DYN), device="cpu",
                       tensor_output=True, max_episode_steps=16, random_kwargs=rk)
    consts = extract_consts(env.envs.dynamics)
    rng = np.random.default_rng(seed + 5)
    q = rng.integers(-127, 128, size=(steps, N, 4), dtype=np.int8)
    actions = decode_actions(q, [-1 / 3, 0, 0, 0], 0.5)
    env.reset()
    imu = [f32(env.sensor_obs["IMU"])]
    state = [f32(env.state)]
    done_count = 0
    for k in range(steps):
        _o, _r, d, _i = env.step(th.from_numpy(actions[k].copy()))
        done_count += int(d.sum())
        imu.append(f32(env.sensor_obs["IMU"]))
        state.append(f32(env.state))
    keep = np.asarray([0, 1, 2, 8, 16, 17, 18, 32, steps], np.int32)      # index 0 = after reset(), k = after step k
    imu, state = np.stack(imu)[keep], np.stack(state)[keep]
    dev = np.abs(imu - state)
    print(f"{name}: N={N} steps={steps} resets={done_count} mean|imu - state|={dev.mean():.4f} max={dev.max():.4f}")
    save = {"kind": np.asarray("hover"), "seed": np.int32(seed), "max_episode_steps": np.int32(16), "actions_q": q,
            "hover": np.asarray([-1 / 3, 0, 0, 0], np.float32), "scale": np.float32(0.5), "imu": imu, "state": state, "keep": keep,
            "noise_mean": np.asarray(IMU_NOISE["IMU"]["kwargs"]["mean"], np.float32),
            "noise_half": np.asarray(IMU_NOISE["IMU"]["kwargs"]["half"], np.float32),
            "label": np.asarray("cr-sqrt-oracle")}
    save.update({"c_" + k: v for k, v in consts.items()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


BPTT_CASES = {
    # name: (env kind, dynamics kwargs, ctor kwargs, hover action, action noise, horizon)
    "bptt_hover_bodyrate": ("hover", ENV_DYN, dict(max_episode_steps=1000), [-1 / 3, 0, 0, 0], 0.3, 12),
    "bptt_hover_thrust": ("hover", RACING_DYN, dict(max_episode_steps=1000), [-0.8333] * 4, 0.05, 12),
    "bptt_hover_resets": ("hover", ENV_DYN, dict(max_episode_steps=5), [-1 / 3, 0, 0, 0], 0.3, 12),
    "bptt_hover_nodelay": ("hover", dict(ENV_DYN, ctrl_delay=False, comm_delay=0.0), dict(max_episode_steps=1000),
                           [-1 / 3, 0, 0, 0], 0.3, 8),
    "bptt_racing_thrust": ("racing", RACING_DYN, dict(max_episode_steps=1000), [-0.8333] * 4, 0.08, 12),
    # RacingEnv2 (envs/RacingEnv.py:218-267): the 16-column gate-relative observation under the loss's obs weights (Wo (H, N, 16))
    "bptt_racing2_thrust": ("racing2", RACING_DYN, dict(max_episode_steps=1000), [-0.8333] * 4, 0.08, 12),
    # repaired RK4: stages chained through the (q, omega) derivatives, acc / tau frozen over the sub-step
    "bptt_hover_rk4": ("hover", dict(ENV_DYN, integrator="rk4"), dict(max_episode_steps=1000), [-1 / 3, 0, 0, 0], 0.3, 12),
    # NavigationEnv reward (progress, view angle through acos, obstacle terms, success bonus): close target so that
    # successes (bonus with a velocity gradient) and their resets fall inside the horizon
    "bptt_nav_bodyrate": ("nav", ENV_DYN, dict(max_episode_steps=1000, target=[1.6, 0., 1.5], random_kwargs={"state_generator": {
        "class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0.5, 1., 0.5]},
                                        "orientation": {"mean": [0., 0., 0.], "half": [0.2, 0.2, 1.0]},
                                        "velocity": {"mean": [1., 0., 0.], "half": [1., .5, .5]}}]}}),
                          [-0.3, 0, 0, 0], 0.5, 12),
    # r05: the observation / reward variants (HoverEnv2: scaled relative-position rows; NavigationEnv2: relative position + the
    # along / across-velocity reward of get_along_vertical_vector) take requires_grad in the reference like every env
    "bptt_hover2": ("hover2", ENV_DYN, dict(max_episode_steps=1000, random_kwargs={"state_generator": {"class": "Uniform", "kwargs": [
        {"position": {"mean": [1., 0., 1.5], "half": [1.0, 1.0, 0.5]}}]}}), [-1 / 3, 0, 0, 0], 0.3, 12),
    "bptt_nav2": ("nav2", ENV_DYN, dict(max_episode_steps=1000, target=[1.6, 0., 1.5], random_kwargs={"state_generator": {
        "class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0.5, 1., 0.5]},
                                        "orientation": {"mean": [0., 0., 0.], "half": [0.2, 0.2, 1.0]},
                                        "velocity": {"mean": [1., 0., 0.], "half": [1., .5, .5]}}]}}),
                  [-0.3, 0, 0, 0], 0.5, 12),
    # velocity / position action types have no BPTT fixture: the reference's autograd raises on them ("one of the variables
    # needed for gradient computation has been modified by an inplace operation": the per-agent loop of dynamics.py:446-450
    # / 481-488 writes pose_err[:, i] / ang_vel_err[:, i] in place while earlier slices are saved for backward), tried
    # with ("hover", dict(ENV_DYN, action_type="velocity"), ..., [0, 0.05, 0, 0], 0.15, 12) and the position analogue
}


def gen_bptt(name, N=64, seed=42):
    """dLoss/dAction through H env steps of the reference with requires_grad=True (torch autograd over
    Dynamics.step + reward; BPTT.py:107-134), loss = sum_t <Wr[t], reward_t> + <Wo[t], obs_t>."""
    HoverEnvShim, NavigationEnv, RacingEnv = import_envs()
    kind, dkw, kw, hover, scale, H = BPTT_CASES[name]
    if dkw.get("integrator") == "rk4":
        repair_rk4("VisFly.utils.maths")
    use_cr_sqrt(True)
    cls = {"hover": HoverEnvShim, "racing": RacingEnv, "nav": NavigationEnv}.get(kind)
    obs_w = 13
    if kind == "racing2":
        cls, kind, obs_w = import_racing2(), "racing", 16       # everything but the observation is RacingEnv
    if cls is None:
        H2, N2 = import_envs2()
        cls = {"hover2": H2, "nav2": N2}[kind]
    kw = dict(kw)
    if "target" in kw:
        kw["target"] = th.tensor(kw["target"])
    env = cls(num_agent_per_scene=N, num_scene=1, seed=seed, visual=False, dynamics_kwargs=dict(dkw), device="cpu",
              requires_grad=True, **({"tensor_output": True} if kind.startswith("hover") else {}), **kw)
    env.tensor_output = True
    if kind == "racing":
        env.targets = th.as_tensor(RACING_TEST_GATES)
    consts = extract_consts(env.envs.dynamics)
    rng = np.random.default_rng(seed + 5)
    acts = th.tensor(decode_actions(rng.integers(-127, 128, size=(H, N, 4), dtype=np.int8), hover, scale),
                     requires_grad=True)
    Wr = th.tensor(rng.normal(size=(H, N)).astype(np.float32))
    Wo = th.tensor((rng.normal(size=(H, N, obs_w)) * 0.1).astype(np.float32))
    env.reset()
    dyn = env.envs.dynamics
    fs_init = f32(dyn.full_state)
    gate0 = env._next_target_i.clone().numpy().astype(np.int32) if kind == "racing" else None
    loss = 0
    dones, rewards, ev_step, ev_agent, ev_fs = [], [], [], [], []
    for t in range(H):
        o, r, d, info = env.step(acts[t])
        loss = loss + (Wr[t] * r).sum() + (Wo[t] * o["state"]).sum()
        dones.append(d.numpy().astype(np.uint8)); rewards.append(f32(r))
        didx = np.nonzero(d.numpy())[0]
        if len(didx):
            fs = f32(dyn.full_state)
            for i in didx:
                ev_step.append(t); ev_agent.append(i); ev_fs.append(fs[i])
    loss.backward()
    g = f32(acts.grad)
    print(f"{name}: N={N} H={H} resets={len(ev_step)} |dL/da| max {np.abs(g).max():.3e} loss {float(loss):.4f}")
    save = {"kind": np.asarray("racing2" if obs_w == 16 else kind), "max_episode_steps": np.int32(kw["max_episode_steps"]), "seed": np.int32(seed),
            "fs_init": fs_init, "actions": f32(acts), "Wr": f32(Wr), "Wo": f32(Wo), "d_actions": g, "loss": np.float64(float(loss)),
            "done": np.stack(dones), "reward": np.stack(rewards),
            "ev_step": np.asarray(ev_step, np.int32), "ev_agent": np.asarray(ev_agent, np.int32),
            "ev_fs": np.stack(ev_fs) if ev_fs else np.zeros((0, 22), np.float32),
            "dyn_kw": np.asarray(repr(dkw)), "label": np.asarray("repaired-oracle" if kind == "racing" or dkw.get("integrator") == "rk4" else "cr-sqrt-oracle"),
            "target": f32(env.target[0]) if kind != "racing" else np.zeros(3, np.float32),
            "spawn": np.asarray(repr(kw.get("random_kwargs", "default")))}
    if kind == "racing":
        save.update(gates=np.asarray(RACING_TEST_GATES, np.float32), gate0=gate0)
    save.update({"c_" + k: v for k, v in consts.items()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def gen_td(name="td_lambda", H=24, N=40, seed=3):
    """compute_td_returns of the reference (utils/algorithms/common.py:893-923), imported through the stubs"""
    import_envs()
    _sb3_stubs()       # the same stand-ins as gen_ppo: utils/algorithms/common.py is imported once per process
    from VisFly.utils.algorithms.common import compute_td_returns
    rng = np.random.default_rng(seed)
    r = rng.normal(size=(H, N)).astype(np.float32)
    nv = rng.normal(size=(H, N)).astype(np.float32)
    done = rng.uniform(size=(H, N)) < 0.12
    ep = done & (rng.uniform(size=(H, N)) < 0.6)
    out = {}
    for tag, e in (("", None), ("_ep", ep)):
        ret = compute_td_returns([th.from_numpy(x) for x in r], [th.from_numpy(x) for x in done], [th.from_numpy(x) for x in nv],
                                 None if e is None else [th.from_numpy(x) for x in e], gamma=0.99, lamda=0.95)
        out["returns" + tag] = np.stack([f32(x) for x in ret])
    print(f"{name}: H={H} N={N} dones={int(done.sum())}")
    np.savez_compressed(os.path.join(OUT, name + ".npz"), r=r, next_value=nv, done=done.astype(np.uint8),
                        episode_done=ep.astype(np.uint8), gamma=np.float64(0.99), lamda=np.float64(0.95), **out)


def _sb3_stubs():
    """the SB3 names utils/algorithms/common.py and utils/policies/extractors.py import, as minimal stand-ins: the code under
    test (RolloutBuffer.compute_returns_and_advantage, StateTargetExtractor, create_mlp) is the reference's own"""
    import torch.nn as nn
    for n in ("stable_baselines3.common.buffers", "stable_baselines3.common.type_aliases", "stable_baselines3.common.preprocessing",
              "stable_baselines3.common.utils", "stable_baselines3.common.vec_env.base_vec_env",
              "stable_baselines3.common.torch_layers", "stable_baselines3.common.policies",
              "stable_baselines3.common.distributions"):
        if n not in sys.modules:
            _auto(n)

    class BaseFeaturesExtractor(nn.Module):          # SB3 torch_layers.BaseFeaturesExtractor: holds features_dim
        def __init__(self, observation_space, features_dim=0):
            super().__init__()
            self._observation_space, self._features_dim = observation_space, features_dim

        @property
        def features_dim(self):
            return self._features_dim

    class BaseBuffer:                                # SB3 buffers.BaseBuffer: sizes + write cursor
        def __init__(self, buffer_size, observation_space, action_space, device="auto", n_envs=1):
            self.buffer_size, self.observation_space, self.action_space = buffer_size, observation_space, action_space
            self.obs_shape, self.action_dim = tuple(observation_space.shape), int(np.prod(action_space.shape))
            self.pos, self.full, self.device, self.n_envs = 0, False, th.device("cpu"), n_envs

        def reset(self):
            self.pos, self.full = 0, False

    sys.modules["stable_baselines3.common.torch_layers"].BaseFeaturesExtractor = BaseFeaturesExtractor
    sys.modules["stable_baselines3.common.buffers"].BaseBuffer = BaseBuffer


def gen_ppo(name, T=16, N=64, seed=11, clip_range_vf=None):
    """PPO-side golden: one rollout T x N + MLP weights -> advantages / returns from the reference's own
    RolloutBuffer.compute_returns_and_advantage (utils/algorithms/common.py:97-132, called the way SB3's collect_rollouts
    calls it: `dones` as a float numpy array), then ONE PPO.train minibatch over all T*N rows: the arithmetic of
    utils/algorithms/PPO.py:210-263 evaluated by torch autograd on a network assembled from the reference's own
    StateTargetExtractor (utils/policies/extractors.py:662-678) and create_mlp (:376-449, what MlpExtractor2 builds,
    policies.py:34-49) plus the action_net / value_net / log_std heads of policies.py:195-254.  Stable-baselines3 itself is
    not installable here; the two SB3 pieces on this path are transcribed in this function and marked [SB3]."""
    import torch.nn as nn
    import_envs()
    _sb3_stubs()
    from VisFly.utils.algorithms.common import RolloutBuffer
    import VisFly.utils.policies.extractors as E
    sp = sys.modules["gymnasium.spaces"]
    rng = np.random.default_rng(seed)
    th.manual_seed(seed)
    gamma, lam, clip_range, ent_coef, vf_coef = 0.99, 0.95, 0.2, 0.01, 0.5
    f = lambda a: th.from_numpy(np.ascontiguousarray(a, np.float32))
    # ---- rollout ----
    obs_state = rng.normal(size=(T, N, 13)).astype(np.float32)
    obs_target = rng.normal(size=(T, N, 3)).astype(np.float32)
    rewards = rng.normal(scale=0.3, size=(T, N)).astype(np.float32)
    values = rng.normal(size=(T, N)).astype(np.float32)
    episode_starts = (rng.uniform(size=(T, N)) < 0.12).astype(np.float32)
    last_values = rng.normal(size=N).astype(np.float32)
    dones = (rng.uniform(size=N) < 0.25).astype(np.float32)
    buf = RolloutBuffer(T, sp.Box(-1, 1, (13,)), sp.Box(-1, 1, (4,)), gae_lambda=lam, gamma=gamma, n_envs=N)
    buf.rewards[:], buf.values[:], buf.episode_starts[:] = f(rewards), f(values), f(episode_starts)
    buf.compute_returns_and_advantage(f(last_values), dones)
    adv, ret = f32(buf.advantages), f32(buf.returns)
    # ---- network (reference modules) ----
    space = sp.Dict({"state": sp.Box(-1, 1, (13,)), "target": sp.Box(-1, 1, (3,))})
    ext = E.StateTargetExtractor(space, net_arch={"state": {"layer": [128, 64]}, "target": {"layer": [128, 64]}}, activation_fn=nn.ReLU)
    pi_net, _ = E.create_mlp(input_dim=128, layer=[64, 64], activation_fn=nn.ReLU)
    vf_net, _ = E.create_mlp(input_dim=128, layer=[64, 64], activation_fn=nn.ReLU)
    action_net, value_net = nn.Linear(64, 4), nn.Linear(64, 1)
    log_std = nn.Parameter(th.full((4,), -0.5))
    with th.no_grad():
        action_net.weight.mul_(0.3)
    linears = [m for net in (ext.state_extractor, ext.target_extractor, pi_net) for m in net if isinstance(m, nn.Linear)]
    linears += [action_net] + [m for m in vf_net if isinstance(m, nn.Linear)] + [value_net]     # visfly_amd MlpPolicy schedule order
    params = [q for m in linears for q in (m.weight, m.bias)] + [log_std]
    flat = np.concatenate([f32(q).reshape(-1) for q in params])
    # ---- the minibatch: all rows, buffer order (swap_and_flatten order is irrelevant for ONE full batch) ----
    M = T * N
    o = {"state": f(obs_state.reshape(M, 13)), "target": f(obs_target.reshape(M, 3))}
    actions = np.tanh(rng.normal(scale=0.8, size=(M, 4))).astype(np.float32)
    A, R, old_v = f(adv.reshape(M)), f(ret.reshape(M)), f(values.reshape(M))
    features = ext.extract(o)                                                        # policies.py:235-236
    mean, v = action_net(pi_net(features)), value_net(vf_net(features)).flatten()    # :240-246
    # [SB3] SquashedDiagGaussianDistribution.log_prob(actions): TanhBijector.inverse + Gaussian log-density - tanh correction
    a = f(actions)
    eps = th.finfo(th.float32).eps
    gauss = 0.5 * (th.log1p(a.clamp(-1 + eps, 1 - eps)) - th.log1p(-a.clamp(-1 + eps, 1 - eps)))
    sd = log_std.exp()
    log_prob = (-((gauss - mean) ** 2) / (2 * sd ** 2) - log_std - np.log(np.sqrt(2 * np.pi))).sum(1)
    log_prob = log_prob - th.log(1 - a ** 2 + 1e-6).sum(1)
    old_log_prob = (log_prob.detach() + f(rng.normal(scale=0.15, size=M))).contiguous()
    # ---- PPO.py:210-263 ----
    advantages = (A - A.mean()) / (A.std() + 1e-8)                                   # :215-220
    ratio = th.exp(log_prob - old_log_prob)                                          # :223
    policy_loss = -th.min(advantages * ratio, advantages * th.clamp(ratio, 1 - clip_range, 1 + clip_range)).mean()   # :226-230
    clip_fraction = th.mean((th.abs(ratio - 1) > clip_range).float())
    values_pred = v if clip_range_vf is None else old_v + th.clamp(v - old_v, -clip_range_vf, clip_range_vf)   # :237-243
    value_loss = th.nn.functional.mse_loss(R, values_pred)                           # :245
    entropy_loss = -th.mean(-log_prob)                                               # :249-251 (squashed Gaussian: no analytic entropy)
    loss = policy_loss + ent_coef * entropy_loss + vf_coef * value_loss              # :257-261
    log_ratio = log_prob - old_log_prob
    approx_kl = th.mean((th.exp(log_ratio) - 1) - log_ratio)                         # :267-270
    loss.backward()
    grad = np.concatenate([f32(q.grad).reshape(-1) for q in params])
    print(f"{name}: T={T} N={N} params={flat.size} loss={float(loss):.6f} pg={float(policy_loss):.6f} v={float(value_loss):.6f} "
          f"kl={float(approx_kl):.5f} clipfrac={float(clip_fraction):.3f} |grad|={np.linalg.norm(grad):.4f}")
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), obs_state=obs_state, obs_target=obs_target, rewards=rewards, values=values,
        episode_starts=episode_starts, last_values=last_values, dones=dones, advantages=adv, returns=ret,
        params=flat, actions=actions, old_log_prob=f32(old_log_prob), adv_normalized=f32(advantages),
        policy_loss=f32(policy_loss), value_loss=f32(value_loss), entropy_loss=f32(entropy_loss), approx_kl=f32(approx_kl),
        clip_fraction=f32(clip_fraction), loss=f32(loss), grad=grad, mean=f32(mean), value=f32(v), log_prob=f32(log_prob),
        gamma=np.float64(gamma), gae_lambda=np.float64(lam), clip_range=np.float32(clip_range), ent_coef=np.float32(ent_coef),
        vf_coef=np.float32(vf_coef), clip_range_vf=np.float32(-1.0 if clip_range_vf is None else clip_range_vf),
        label=np.asarray("reference RolloutBuffer + StateTargetExtractor + create_mlp; SB3 distribution transcribed"))


def gen_ckpt_keys(name="ckpt_keys"):
    """parameter names and shapes of the reference's own policy sub-modules for the default StateTarget network: the
    StateTargetExtractor instance (utils/policies/extractors.py:662-678; sub-module names come from set_mlp_feature_extractor,
    :464-486) and the two create_mlp trunks MlpExtractor2 holds as policy_net / value_net (policies.py:34-49).  Pins the key
    names visfly_amd/checkpoint.py writes below SB3's own prefixes (features_extractor. / mlp_extractor.*)."""
    import json
    import torch.nn as nn
    import_envs()
    _sb3_stubs()
    import VisFly.utils.policies.extractors as E
    sp = sys.modules["gymnasium.spaces"]
    space = sp.Dict({"state": sp.Box(-1, 1, (13,)), "target": sp.Box(-1, 1, (3,))})
    ext = E.StateTargetExtractor(space, net_arch={"state": {"layer": [128, 64]}, "target": {"layer": [96, 32]}}, activation_fn=nn.ReLU)
    pi_net, _ = E.create_mlp(input_dim=96, layer=[64, 48], activation_fn=nn.ReLU)
    vf_net, _ = E.create_mlp(input_dim=96, layer=[64, 64], activation_fn=nn.ReLU)
    keys = {}
    for prefix, mod in (("features_extractor", ext), ("mlp_extractor.policy_net", pi_net), ("mlp_extractor.value_net", vf_net)):
        for k, v in mod.state_dict().items():
            keys[f"{prefix}.{k}"] = list(v.shape)
    keys["features_dim"] = int(ext.features_dim)
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(keys, f, indent=1, sort_keys=True)
    print(f"{name}: {len(keys) - 1} parameter tensors, features_dim {keys['features_dim']}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    for name in DYN_CASES:
        if args.only in (None, name):
            gen_dyn(name)
    for name in ENV_CASES:
        if args.only in (None, name):
            gen_env(name)
    for name in BPTT_CASES:
        if args.only in (None, name):
            gen_bptt(name)
    if args.only in (None, "td_lambda"):
        gen_td()
    if args.only in (None, "ckpt_keys"):
        gen_ckpt_keys()
    if args.only in (None, "env_hover_imu"):
        gen_imu()
    if args.only in (None, "ppo_nav"):
        gen_ppo("ppo_nav")
    if args.only in (None, "ppo_nav_vclip"):
        gen_ppo("ppo_nav_vclip", seed=12, clip_range_vf=0.3)


if __name__ == "__main__":
    main()
