"""``Dynamics`` -- drop-in for the reference's batched quadrotor dynamics
(envs/base/dynamics.py:19) whose ``step``/``reset`` run as single fused HIP launches.

Same constructor kwargs (``dynamics_kwargs``), same ``reset``/``step`` signatures and
return shapes, same properties.  State lives in ONE device slab ``[N/64][G][64][4]`` fp32 (AoSoA,
tiled per wavefront in 16-byte granules, include/visfly_amd.h); the public properties
gather the requested components out of it.
"""
import math
from typing import List, Optional, Tuple, Union

import numpy as np
import torch as th

from . import _lib
from ._lib import G_ACC, G_AACC, G_MOT, G_OMG, G_POS, G_QUAT, G_THR, G_VEL, TILE, VisflyError
from .constants import ACTION_TYPES, derive_constants


class ACTION_TYPE:
    """value-compatible with the reference enum (utils/type.py:14-18)"""
    THRUST, BODYRATE, VELOCITY, POSITION = 0, 1, 2, 3


def _as_device_f32(x, device, cols=None):
    if x is None:
        return None
    t = th.as_tensor(x, dtype=th.float32).to(device, non_blocking=True)
    if cols is not None:
        t = t.reshape(-1, cols)
    return t.contiguous()


class Dynamics:
    def __init__(
            self,
            num: int = 1,
            action_type: str = "bodyrate",
            ori_output_type: str = "quaternion",
            seed: int = 42,
            dt: float = 0.005,
            ctrl_dt: float = 0.03,
            ctrl_delay: bool = True,
            comm_delay: float = 0.06,
            action_space: Tuple[float, float] = (-1, 1),
            device: Union[str, th.device] = "cuda",
            integrator: str = "euler",
            drag_random: float = 0,
            cfg: Union[str, dict] = "drone_state",
            wind_settings: Optional[List] = (0, 0, 0),
            rotor_sim: bool = True,
            transcendentals: str = "cr",
            constants: Optional[dict] = None,
            _attach=None,
    ):
        assert action_type in ["bodyrate", "thrust", "velocity", "position"]
        assert ori_output_type in ["quaternion", "euler"]
        self.device = th.device(device)
        if self.device.type != "cuda":
            raise VisflyError(
                f"visfly_amd.Dynamics runs on an MI355X only (device='{device}'); there is no CPU fallback")
        if self.device.index is None:
            self.device = th.device("cuda", th.cuda.current_device())
        self.num = int(num)
        self.action_type = ACTION_TYPES[action_type]
        self.angular_output_type = ori_output_type
        self._is_quat_output = ori_output_type == "quaternion"
        self.dt, self.ctrl_dt = dt, ctrl_dt
        self._integrator = integrator
        self._ctrl_delay = ctrl_delay
        self._drag_random = drag_random
        self._rotor_sim = rotor_sim
        # `constants` lets parity tests inject the golden fixture's constant bits verbatim
        self.constants = dict(constants) if constants is not None else derive_constants(
            action_type=action_type, dt=dt, ctrl_dt=ctrl_dt, ctrl_delay=ctrl_delay, comm_delay=comm_delay,
            action_space=action_space, integrator=integrator, cfg=cfg, wind_settings=wind_settings,
            transcendentals=transcendentals)
        c = self.constants
        self._interval_steps = int(c["interval_steps"])
        self._comm_delay_steps = int(c["delay_steps"])
        self.m = th.tensor(float(c["m"]))
        self.name = cfg if isinstance(cfg, str) else cfg.get("name", "custom")

        self.set_seed(seed)
        self._owns_handle = True
        self._wind_const = th.as_tensor(np.asarray(c["wind"], np.float32), device=self.device).reshape(1, 3)
        self._wind_fn = None
        if _attach is None:
            with th.cuda.device(self.device):
                self._cfg = _lib.DynCfg.from_dict(c)
                h = _lib._vp()
                _lib.check(_lib.lib().vf_dyn_create(self._cfg, self.num, 1 if drag_random else 0, h))
                self._h = h
                self._G = int(_lib.lib().vf_dyn_granules(self._h))
                floats = int(_lib.lib().vf_dyn_slab_floats(self._h))
                self._slab = th.zeros((floats // (self._G * TILE * 4), self._G, TILE, 4), dtype=th.float32,
                                      device=self.device)
                _lib.check(_lib.lib().vf_dyn_bind(self._h, _lib.ptr(self._slab)))
            self.reset()
        else:  # embedded in an env handle that owns slab and lifetime (DroneEnvsBase.dynamics)
            self._h, self._slab, self._G = _attach
            self._cfg = _lib.DynCfg.from_dict(c)
            self._owns_handle = False
        if len(wind_settings) and isinstance(wind_settings[0], str):
            self._create_wind(wind_settings)

    # ------------------------------------------------------------------ wind functions (dynamics.py:132-174,384-388)
    def _create_wind(self, wind_settings):
        """six strings = two (x, y, z) triples of expressions in x (= the agents' time, (N,)) and y (= the previous value of
        that component, (N,)), eval'ed into lambdas exactly as the reference does -- arbitrary Python from the caller's config,
        like there.  The reference's three-string form builds a 3-argument lambda and calls it with two (:159-165): it raises
        TypeError inside its own constructor, so there is nothing to mirror."""
        if len(wind_settings) == 3:
            raise NotImplementedError("wind_settings with three strings cannot run in the reference (dynamics.py:159-165 builds "
                                      "`lambda x,y,z` and update_wind calls it with two arguments); pass six strings")
        if len(wind_settings) != 6:
            raise ValueError("wind_settings should be a list of length 3 or 6, or a string function")           # :168
        ns = {"th": th, "torch": th, "np": np, "math": math}
        fn = [eval("lambda x,y:" + s, ns) for s in wind_settings]                                                 # :139-144
        self._wind_fn = (fn[:3], fn[3:])
        self._wind_1 = th.zeros((3, self.num), device=self.device)                                                # :172-173
        self._wind_2 = th.zeros((3, self.num), device=self.device)
        self._wind_rows = th.zeros((self.num, 4), device=self.device)
        self.update_wind()                                                                                        # :174
        _lib.check(_lib.lib().vf_dyn_set_wind(self._h, _lib.ptr(self._wind_rows)))

    def update_wind(self):
        """dynamics.py:384-388; runs at the top of every step (:320).  The result goes to the per-agent rows the kernels read."""
        if self._wind_fn is None:
            return
        t = self.t
        f1, f2 = self._wind_fn
        self._wind_1 = th.stack([f1[0](t, self._wind_1[0]), f1[1](t, self._wind_1[1]), f1[2](t, self._wind_1[2])])
        self._wind_2 = th.stack([f2[0](t, self._wind_2[0]), f2[1](t, self._wind_2[1]), f2[2](t, self._wind_2[2])])
        self._wind_rows[:, :3] = (self._wind_1 + self._wind_2).T

    @property
    def _wind(self):
        """(1,3) constant wind or (N,3) per-agent rows"""
        return self._wind_const if self._wind_fn is None else self._wind_rows[:, :3]
    # ------------------------------------------------------------------ lifecycle
    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and getattr(self, "_owns_handle", False):
            _lib.lib().vf_dyn_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_seed(self, seed=42):
        """reference: th.manual_seed(seed) on the global CPU generator (dynamics.py:556-557);
        here the stream lives in ``self.rng`` (shared with the env layer for replay parity)."""
        self.rng = th.Generator(device="cpu")
        self.rng.manual_seed(int(seed))

    def detach(self):
        """state tensors carry no autograd graph on the kernel path (dynamics.py:176-190)"""
        return None

    def _stream(self):
        return _lib.current_stream(self.device)

    def _drag_rows(self, f_lin, f_quad, k):
        """k = k_mean * factor with factor (3,) shared or (k,3) per agent (dynamics.py:244-246) -> (k,3) x2"""
        c = self.constants
        kl = th.as_tensor(np.asarray(c["k_lin"], np.float32)).reshape(1, 3) * f_lin
        kq = th.as_tensor(np.asarray(c["k_quad"], np.float32)).reshape(1, 3) * f_quad
        return kl.expand(k, 3).contiguous(), kq.expand(k, 3).contiguous()

    # ------------------------------------------------------------------ reset / step
    def reset(
            self,
            pos=None, ori=None, vel=None, ori_vel=None, motor_omega=None, thrusts=None, t=None,
            indices: Optional[List] = None,
            t_rand=None,
    ):
        """Dynamics.reset (dynamics.py:218-269): full reset (indices None) or scatter into
        ``indices``.  ``t_rand``: optional uniform draws for the indexed ``t <- U*6.28``; by
        default they come from ``self.rng`` in the reference's order."""
        dev = self.device
        with th.cuda.device(dev):
            idx = None
            k = self.num
            klin = kquad = None
            r = self._drag_random
            if indices is not None:
                idx = th.as_tensor(indices, dtype=th.int32).reshape(-1).to(dev, non_blocking=True).contiguous()
                k = idx.numel()
                if t is None and t_rand is None:
                    t_rand = th.rand((k,), generator=self.rng)
                if r:  # per-agent redraw (the reference raises IndexError here, SURVEY C-2)
                    fl = ((th.rand((k, 3), generator=self.rng) - 0.5) * 2 * r).clamp(-0.5, .5) + 1
                    fq = ((th.rand((k, 3), generator=self.rng) - 0.5) * 2 * r).clamp(-0.5, .5) + 1
                    klin, kquad = self._drag_rows(fl, fq, k)
            elif r:  # one (3,1) factor pair shared by all agents, like the reference's full reset
                fl = ((th.rand((3, 1), generator=self.rng) - 0.5) * 2 * r).clamp(-0.5, .5) + 1
                fq = ((th.rand((3, 1), generator=self.rng) - 0.5) * 2 * r).clamp(-0.5, .5) + 1
                klin, kquad = self._drag_rows(fl.reshape(1, 3), fq.reshape(1, 3), k)
            args = [_as_device_f32(pos, dev, 3), _as_device_f32(ori, dev, 4), _as_device_f32(vel, dev, 3),
                    _as_device_f32(ori_vel, dev, 3), _as_device_f32(motor_omega, dev, 4),
                    _as_device_f32(thrusts, dev, 4), _as_device_f32(t, dev), _as_device_f32(t_rand, dev),
                    _as_device_f32(klin, dev, 3), _as_device_f32(kquad, dev, 3)]
            for a in args:
                if a is not None and a.shape[0] != k:
                    raise ValueError(f"reset: expected {k} rows, got {tuple(a.shape)}")
            _lib.check(_lib.lib().vf_dyn_reset(self._h, _lib.ptr(idx), k, *[_lib.ptr(a) for a in args],
                                               self._stream()))
            self._keepalive = (idx, args)  # until the stream has consumed them
        return self.state

    def step(self, action) -> th.Tensor:
        """One control interval (dynamics.py:319-372) -> state (N,13)."""
        with th.cuda.device(self.device):
            a = _as_device_f32(action, self.device, 4)
            if a.shape[0] != self.num:
                raise ValueError(f"step: action must be ({self.num},4), got {tuple(a.shape)}")
            out = th.empty((self.num, 13), dtype=th.float32, device=self.device)
            self.update_wind()                                                              # :320
            _lib.check(_lib.lib().vf_dyn_step(self._h, _lib.ptr(a), _lib.ptr(out), self._stream()))
            self._last_action = a
        return out

    def backward_step(self, tape_slab, action, d_state, adj_slab):
        """reverse pass of ONE step() of a bare Dynamics object (vf_dyn_step_bwd; what autograd does through
        dynamics.py:319-372 when the loss reads the returned state): ``tape_slab`` = a clone of ``self._slab`` taken before
        that step, ``action`` what it was given, ``d_state`` (N,13) dLoss/d(returned state) or None, ``adj_slab`` the adjoint
        of the persistent state (same shape as the slab, in/out: zeros after the last step of the horizon).  -> dLoss/d action"""
        with th.cuda.device(self.device):
            a = _as_device_f32(action, self.device, 4)
            d_action = th.empty((self.num, 4), dtype=th.float32, device=self.device)
            ds = None if d_state is None else d_state.to(self.device, dtype=th.float32).reshape(self.num, 13).contiguous()
            _lib.check(_lib.lib().vf_dyn_step_bwd(self._h, _lib.ptr(tape_slab), _lib.ptr(a), _lib.ptr(ds), _lib.ptr(adj_slab),
                                                  _lib.ptr(d_action), self._stream()))
        return d_action

    # ------------------------------------------------------------------ properties
    def _vec(self, g):
        """(N,3) copy of the xyz components of granule g"""
        return self._slab[:, g, :, 1:4].reshape(-1, 3)[:self.num]

    def _gran(self, g):
        """(N,4) copy of granule g"""
        return self._slab[:, g].reshape(-1, 4)[:self.num]

    @property
    def position(self):
        return self._vec(G_POS)

    @property
    def quaternion(self):
        return self._gran(G_QUAT)

    @property
    def orientation(self):
        if self._is_quat_output:
            return self._gran(G_QUAT)
        return self._euler().T

    def _euler(self):
        w, x, y, z = self._gran(G_QUAT).T
        roll = th.atan2(2 * (w * x + y * z), 1 - 2 * (x.pow(2) + y.pow(2)))       # maths.py:244-249
        pitch = th.asin(2 * (w * y - z * x))
        yaw = th.atan2(2 * (w * z + x * y), 1 - 2 * (y.pow(2) + z.pow(2)))
        return th.stack([roll, pitch, yaw])

    @property
    def direction(self):
        w, x, y, z = self._gran(G_QUAT).T                                          # maths.py:123-133
        return th.stack([1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w)]).T

    @property
    def R(self):
        w, x, y, z = self._gran(G_QUAT).T                                          # maths.py:110-120
        return th.stack([
            th.stack([1 - 2 * (y.pow(2) + z.pow(2)), 2 * (x * y - z * w), 2 * (x * z + y * w)]),
            th.stack([2 * (x * y + z * w), 1 - 2 * (x.pow(2) + z.pow(2)), 2 * (y * z - x * w)]),
            th.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x.pow(2) + y.pow(2))])])

    @property
    def xz_axis(self):
        """rows 0 and 2 of ``R`` as the reference's Quaternion.xz_axis returns them, (2, 3, N) (dynamics.py:824-826, maths.py:134-151)"""
        w, x, y, z = self._gran(G_QUAT).T
        return th.stack([
            th.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)]),
            th.stack([2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)])])

    @property
    def velocity(self):
        return self._vec(G_VEL) + self._wind

    @property
    def angular_velocity(self):
        return self._vec(G_OMG)

    @property
    def acceleration(self):
        return self._vec(G_ACC)

    @property
    def angular_acceleration(self):
        return self._vec(G_AACC)

    @property
    def t(self):
        return self._slab[:, G_POS, :, 0].reshape(-1)[:self.num]

    @property
    def motor_omega(self):
        return self._gran(G_MOT)

    @property
    def thrusts(self):
        return self._gran(G_THR)

    @property
    def drag_coefficients(self):
        """per-agent (N,3) linear and quadratic drag coefficients, or None when they are shared constants"""
        if not self._drag_random:
            return None
        g = _lib.G_RING + self._comm_delay_steps
        return self._vec(g), self._vec(g + 1)

    @property
    def delay_ring(self):
        """(D, N, 4) delayed actions in slot order (slot = ring head at the time of the push)"""
        D = self._comm_delay_steps
        return th.stack([self._gran(_lib.G_RING + s) for s in range(D)]) if D else None

    @property
    def wind_velocity(self):
        return self._wind.T.expand(3, self.num)                 # (3,N) as the reference's attribute

    @property
    def is_quat_output(self):
        return self._is_quat_output

    @property
    def state(self):
        return th.hstack([self.position, self.orientation, self.velocity, self.angular_velocity])

    @property
    def full_state(self):
        return th.hstack([self.position, self.orientation, self.velocity, self.angular_velocity,
                          self.motor_omega, self.thrusts, self.t.unsqueeze(1)])

    @property
    def extend_state(self):
        return th.hstack([self.position, self.orientation, self.velocity, self.angular_velocity,
                          self.acceleration, self.angular_acceleration, self.motor_omega, self.thrusts,
                          self.t.unsqueeze(1)])

    # ------------------------------------------------------------------ measurement helper
    def time_steps(self, action, iters=100):
        """mean device microseconds per fused-step launch, HIP events on the current stream"""
        import ctypes
        a = _as_device_f32(action, self.device, 4)
        out = th.empty((self.num, 13), dtype=th.float32, device=self.device)
        us = ctypes.c_float(0)
        _lib.check(_lib.lib().vf_dyn_time_steps(self._h, _lib.ptr(a), _lib.ptr(out), int(iters), self._stream(),
                                                ctypes.byref(us)))
        return float(us.value)
