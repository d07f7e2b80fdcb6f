"""VecEnv surface of the visual=False drone envs, backed by the fused HIP env-step kernel.

Mirrors ``DroneGymEnvsBase`` / ``DroneEnvsBase`` of the reference (envs/base/droneGymEnv.py:19,
envs/base/droneEnv.py:18): same constructor kwargs, ``reset() / step(action[, is_test]) /
reset_agent_by_id() / get_observation() / detach()``, same ``(obs, reward, done, info)``
conventions (obs = post-auto-reset, reward/done = pre-reset; ``tensor_output`` switches the
numpy return path, droneGymEnv.py:209-218), same properties.

Two spawn modes:
  ``spawn="device"`` (default): done agents are re-spawned inside the step kernel with a Philox
      counter RNG -- one launch per step, no host round trip.
  ``spawn="replay"``: parity mode.  The reference's global-RNG draw order is
      replayed on the host generator shared with ``Dynamics`` and the drawn states are scattered
      by ``vf_env_reset``; reset states, counters and done flags are then bit-identical to the
      reference on the same seed (one host sync per step).  The spawned orientation goes through ``Quaternion.from_euler``
      (sin / cos): ``replay_trig="torch"`` (default) evaluates it with this torch build's own CPU sin / cos, i.e. draws what the
      reference AS TORCH RUNS IT draws; ``replay_trig="cr"`` evaluates it the way the golden generator's CR-trig patch does (fp64
      result rounded once) and is what the tests pass to reproduce the CR-patched fixtures (1 ulp apart on a few per cent of the
      arguments).
"""
import ctypes as C
import os
from typing import Dict, List, Optional

import numpy as np
import torch as th

from .. import _lib
from .._lib import TILE, VisflyError
from ..constants import derive_constants
from ..dynamics import Dynamics
from ..type import TensorDict
from . import spaces
from .randomization import ReplaySpawner, spawn_boxes

# hot-path helpers: the raw handle of torch's current stream / device without building Python wrapper objects
# (the env step is one ~11 us launch; the host side of a step must stay below that)
_raw_stream = getattr(th._C, "_cuda_getCurrentRawStream", None) or (lambda idx: th.cuda.current_stream(idx).cuda_stream)
_cuda_get_device = getattr(th._C, "_cuda_getDevice", None) or th.cuda.current_device

HOVER, NAV, RACING = 0, 1, 2
F_EPISODE_DONE, F_ONCE_COLLIDED, F_COLLISION, F_OUT_BOUNDS, F_SUCCESS, F_FAILURE, F_DONE = 1, 2, 4, 8, 16, 32, 64
EP_SUCCESS, EP_TRUNCATED, EP_COLLIDED, EP_EPISODE_DONE = 1, 2, 4, 8


class _Info(list):
    """per-agent info dicts (droneGymEnv.py:238-275), materialised from the step's device outputs
    on first access so that the hot loop never syncs for them.  The episode buffers hold, per agent,
    the most recently finished episode (what the reference keeps in ``self._info[i]``); read an info
    before that agent finishes another episode."""

    def __init__(self, env, done, ep_return, ep_length, ep_flags, terminal_obs, extra=None):
        super().__init__()
        self._src = (env, done, ep_return, ep_length, ep_flags, terminal_obs, extra)
        self._ready = False

    def _build(self):
        if self._ready:
            return
        env, done, ret, length, flags, tobs, extra = self._src
        self._ready = True
        super().extend(env._info)
        idx = th.nonzero(done).flatten().cpu().numpy()
        if len(idx):
            ret, length, flags = ret.cpu().numpy(), length.cpu().numpy(), flags.cpu().numpy()
            tobs = tobs.cpu() if not env.tensor_output else tobs
            for i in idx:
                f = int(flags[i])
                d = {"episode_done": bool(f & EP_EPISODE_DONE), "is_success": bool(f & EP_SUCCESS),
                     "episode": {"r": np.asarray(ret[i]), "l": np.asarray(length[i]),
                                 "t": np.asarray(length[i] * np.float32(env.envs.dynamics.ctrl_dt)),
                                 "extra": {"collision": np.asarray(bool(f & EP_COLLIDED))}},
                     "terminal_observation": {"state": env._terminal_state_obs(tobs, i), **env._terminal_static_obs(i)},
                     "TimeLimit.truncated": bool(f & EP_TRUNCATED)}
                if extra is not None:
                    d["episode"]["extra"].update({k: v[i].item() for k, v in extra.items()})
                super().__setitem__(int(i), d)
            env._info = list(super().__iter__())

    def _clear(self):
        """back to the un-materialised state (ring slots re-use their info object)"""
        super().clear()
        self._ready = False

    def __getitem__(self, i):
        self._build()
        return super().__getitem__(i)

    def __iter__(self):
        self._build()
        return super().__iter__()

    def __len__(self):
        return self._src[0].num_agent

    def copy(self):
        self._build()
        return list(super().__iter__())


class _OutSlot:
    """one pre-allocated output set of step(): tensors, the observation dict, the lazy info list and the C struct
    that points at them (out_buffers > 0)"""
    __slots__ = ("state", "reward", "done", "obs", "info", "outs", "ref")

    def __init__(self, env):
        N, dev = env.num_agent, env.device
        self.state = th.zeros((N, 13), dtype=th.float32, device=dev)
        self.reward = th.zeros(N, dtype=th.float32, device=dev)
        self.done = th.zeros(N, dtype=th.bool, device=dev)
        self.outs = env._out(self.state, self.reward, self.done)
        self.ref = C.byref(self.outs)
        self.obs = None
        self.info = _Info(env, self.done, env._ep_return, env._ep_length, env._ep_flags, env._terminal_obs, None)


def _imu_noise_model(random_kwargs):
    """random_kwargs["noise_kwargs"]["IMU"] (droneEnv.py:53,99-112) -> None | (mean(13,), half(13,)) of the uniform model"""
    nk = (random_kwargs or {}).get("noise_kwargs") or {}
    unknown = set(nk) - {"IMU"}
    if unknown:
        raise NotImplementedError(f"noise_kwargs for {sorted(unknown)}: only the IMU (state) noise exists with visual=False")
    imu = nk.get("IMU")
    if imu is None:
        return None
    model = imu.get("model", "UniformNoiseModel")
    if model == "GaussianNoiseModel":
        raise NotImplementedError("GaussianNoiseModel: the reference's Normal.generate (utils/type.py:56-57) calls "
                                  "th.normal(mean, std, int) and raises for every input, so there is nothing to mirror")
    if model != "UniformNoiseModel":
        raise ValueError("IMU Noise model does not exist.")                                   # droneEnv.py:112
    kw = imu.get("kwargs", {})
    mean = th.atleast_1d(th.as_tensor(kw.get("mean", th.zeros(13)), dtype=th.float32))
    half = th.atleast_1d(th.as_tensor(kw.get("half", th.zeros(13)), dtype=th.float32))
    if mean.numel() != 13 or half.numel() != 13:
        raise ValueError("IMU noise mean / half must have 13 entries (one per state component)")
    return mean, half


def _reject_unsupported(scene_kwargs, sensor_kwargs):
    """With visual=False the reference never renders (droneEnv.py:296-333 reads the sensors only if self.visual) and forces the
    empty box scene (:69-71), so visual sensors / scene objects change nothing there either -- say so once instead of
    dropping them silently."""
    import warnings
    visual = [s.get("uuid") for s in (sensor_kwargs or []) if "IMU" not in str(s.get("uuid", ""))]
    if visual:
        warnings.warn(f"visfly_amd: sensor_kwargs {visual} are not rendered (visual=False path; the reference ignores them "
                      "too when visual=False)", stacklevel=4)
    ignored = [k for k in ("obj_settings", "render_settings", "update_approaching_info", "path") if (scene_kwargs or {}).get(k)]
    if ignored:
        warnings.warn(f"visfly_amd: scene_kwargs {ignored} have no effect with visual=False (collisions use the "
                      "[-30,-30,0]..[30,30,8] box of droneEnv.py:129)", stacklevel=4)


class DroneEnvsBase:
    """``env.envs`` of the reference (envs/base/droneEnv.py:18): owns ``dynamics`` and the
    collision queries; here a view onto the env's device handle."""

    def __init__(self, owner, dynamics: Dynamics):
        self._o = owner
        self.dynamics = dynamics
        self.device = dynamics.device
        self.visual = False
        self.uav_radius = 0.1
        self.sceneManager = None

    def _q(self, name):
        return self._o._query()[name]

    is_collision = property(lambda s: (s._q("flags") & F_COLLISION) != 0)
    is_out_bounds = property(lambda s: (s._q("flags") & F_OUT_BOUNDS) != 0)
    once_collided = property(lambda s: (s._q("flags") & F_ONCE_COLLIDED) != 0)
    collision_point = property(lambda s: s._q("col_point"))
    collision_vector = property(lambda s: s._q("col_vec"))
    collision_dis = property(lambda s: s._q("col_dis"))

    def __getattr__(self, name):  # state, position, orientation, velocity, ... (droneEnv.py:416-478)
        if name in ("state", "position", "orientation", "velocity", "angular_velocity", "direction", "t",
                    "thrusts", "full_state", "extend_state", "acceleration", "angular_acceleration"):
            return getattr(self.dynamics, name)
        raise AttributeError(name)

    def detach(self):
        self.dynamics.detach()

    def stack(self):
        """droneEnv.py:387-393: in-memory snapshot of the pose of every agent"""
        d = self.dynamics
        self._stack_cache = (d.position.clone(), d.quaternion.clone(), d.velocity.clone(), d.angular_velocity.clone())

    def recover(self):
        """droneEnv.py:395-396: reset_agents(state=snapshot) -- positions are re-drawn (pos_reset_by_state=False)"""
        p, q, v, w = self._stack_cache
        o, d = self._o, self.dynamics
        fs = th.zeros((d.num, 22), device=d.device)
        fs[:, 3:7], fs[:, 7:10], fs[:, 10:13] = q, v - d._wind, w
        fs[:, 13:17], fs[:, 17:21] = float(d.constants["w_init"]), float(d.constants["T_init"])
        if o.spawn_mode == "replay":
            fs[:, 0:3] = o._spawner.generate(d.num)[0].to(d.device)
        else:
            o._reset_kernel(None, None)               # device spawn draws the positions ...
            fs[:, 0:3] = d.position
        o._reset_kernel(None, fs)                     # ... then the snapshot attitude / velocities are restored

    def close(self):
        pass


class DroneGymEnvsBase:
    KIND = HOVER
    OBS_MODE = 0        # VF_OBS_STATE
    _OBS_W = 13         # columns of the "state" rows the kernels write (RacingEnv2's persistent launches: 16)
    REWARD_MODE = 0     # VF_REWARD_DEFAULT
    _STATIC_OBS_CONST = True   # _static_obs() returns the same tensor objects every step (the ring path caches the dict)

    def __init__(
            self,
            num_agent_per_scene: int = 1,
            num_scene: int = 1,
            seed: int = 42,
            visual: bool = False,
            max_episode_steps: int = 1000,
            device="cuda",
            dynamics_kwargs: Optional[Dict] = None,
            random_kwargs: Optional[Dict] = None,
            requires_grad: bool = False,
            scene_kwargs: Optional[Dict] = None,
            sensor_kwargs: Optional[List] = None,
            tensor_output: bool = True,
            is_train: bool = False,
            is_collision_reset: bool = True,
            spawn: str = "device",
            validate_actions: Optional[bool] = None,
            target=None,
            success_radius: float = 0.5,
            gates=None,
            constants: Optional[dict] = None,
            out_buffers: int = 0,
            spawn_prefetch: Optional[bool] = None,
            replay_trig: str = "torch",
    ):
        """out_buffers = R > 0: step() writes into a ring of R pre-allocated (obs, reward, done) sets instead of fresh tensors
        -- no allocation and no Python object construction on the hot path; what step t returned stays valid until step
        t + R.  0 (default) returns fresh tensors every step like the reference.
        spawn_prefetch (default: on for spawn="device" without requires_grad): the state an agent re-spawns into is drawn ahead of
        its episode end by helper blocks of the step launch and kept in the slab (8 granules per agent; include/visfly_amd.h
        "Prefetched re-spawn") -- bit-identical results, the re-spawning wave no longer holds up the launch."""
        if visual:
            raise NotImplementedError("visual=True needs the external Habitat-sim renderer; the MI355X engine "
                                      "covers the visual=False path (SURVEY.md 8)")
        if spawn not in ("device", "replay"):
            raise ValueError("spawn must be 'device' or 'replay'")
        if replay_trig not in ("torch", "cr"):
            raise ValueError("replay_trig must be 'torch' or 'cr'")
        self.device = th.device(device)
        if self.device.type != "cuda":
            raise VisflyError(f"visfly_amd envs run on an MI355X only (device='{device}'); there is no CPU fallback")
        if self.device.index is None:
            self.device = th.device("cuda", th.cuda.current_device())
        dynamics_kwargs = dict(dynamics_kwargs or {})
        self.num_agent = self.num_envs = num_agent_per_scene * num_scene
        self.num_scene, self.num_agent_per_scene = num_scene, num_agent_per_scene
        self.requires_grad, self.tensor_output = requires_grad, tensor_output
        self.is_train, self.is_collision_reset = is_train, is_collision_reset
        self.max_episode_steps = int(max_episode_steps)
        self.max_sense_radius = 10
        self.spawn_mode = spawn
        # persistent BPTT roll-outs also record the integrator sub-steps for the reverse launch (VISFLY_AMD_SUBSTEP_TAPE=0: A/B switch)
        self.substep_tape = os.environ.get("VISFLY_AMD_SUBSTEP_TAPE", "1") != "0"
        self.validate_actions = (spawn == "replay") if validate_actions is None else validate_actions
        self.seed = seed
        N = self.num_agent

        dkw = {k: v for k, v in dynamics_kwargs.items() if k not in ("seed", "device", "num")}
        drag_random = dkw.get("drag_random", 0)
        consts = constants if constants is not None else derive_constants(
            **{k: v for k, v in dkw.items() if k in ("action_type", "dt", "ctrl_dt", "ctrl_delay", "comm_delay",
                                                     "action_space", "integrator", "cfg", "wind_settings", "transcendentals")})
        self._boxes = spawn_boxes(random_kwargs)
        self._imu_noise = _imu_noise_model(random_kwargs)
        _reject_unsupported(scene_kwargs, sensor_kwargs)     # warns
        self.target = th.as_tensor([1., 0., 1.5] if target is None else target, dtype=th.float32).reshape(3)
        self.success_radius = success_radius
        self.targets = None if gates is None else th.as_tensor(gates, dtype=th.float32)

        e = _lib.EnvCfg()
        e.kind, e.max_episode_steps = self.KIND, self.max_episode_steps
        e.obs_mode, e.reward_mode = self.OBS_MODE, self.REWARD_MODE
        e.sense_radius = 10.0                                                         # droneGymEnv.py:69 (max_sense_radius; VF_OBS_RACE2)
        e.is_collision_reset = int(bool(is_collision_reset))
        lo, hi = (-30., -30., 0.), (30., 30., 8.)                                     # droneEnv.py:129
        for d in range(3):
            e.bbox_lo[d], e.bbox_hi[d], e.target[d] = lo[d], hi[d], float(self.target[d])
        e.uav_radius, e.success_radius = 0.1, float(success_radius)
        e.n_gates = 0 if gates is None else len(gates)
        for gi in range(e.n_gates):
            for d in range(3):
                e.gates[gi][d] = float(gates[gi][d])
        if len(self._boxes) > _lib.MAX_SPAWN:
            raise ValueError(f"at most {_lib.MAX_SPAWN} spawn boxes")
        e.n_spawn = len(self._boxes)
        e.drag_random = float(drag_random or 0.0)
        self._drag_random = float(drag_random or 0.0)
        for bi, b in enumerate(self._boxes):
            sb = e.spawn[bi]
            for name, f in (("pos", "position"), ("ori", "orientation"), ("vel", "velocity"), ("omg", "angular_velocity")):
                for d in range(3):
                    getattr(sb, name + "_mean")[d] = b[f]["mean"][d]
                    getattr(sb, name + "_half")[d] = b[f]["half"][d]
        e.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        if spawn_prefetch is None:       # the tape of a requires_grad env copies the slab every step: keep it small there
            spawn_prefetch = spawn == "device" and not requires_grad
        e.spawn_prefetch = int(bool(spawn_prefetch))
        self._ecfg = e

        L = _lib.lib()
        with th.cuda.device(self.device):
            self._dcfg = _lib.DynCfg.from_dict(consts)
            h = _lib._vp()
            _lib.check(L.vf_env_create(self._dcfg, self._ecfg, N, 1 if drag_random else 0, h))
            self._h = h
            G = int(L.vf_env_granules(h))
            floats = int(L.vf_env_slab_floats(h))
            self._slab = th.zeros((floats // (G * TILE * 4), G, TILE, 4), dtype=th.float32, device=self.device)
            _lib.check(L.vf_env_bind(h, _lib.ptr(self._slab)))
            hd = _lib._vp(L.vf_env_dyn(h))
            dyn = Dynamics(num=N, seed=seed, device=self.device, constants=consts,
                           **{k: v for k, v in dkw.items() if k != "constants"}, _attach=(hd, self._slab, G))
            self.envs = DroneEnvsBase(self, dyn)
            self._spawner = ReplaySpawner(self._boxes, dyn.rng, cr_trig=replay_trig == "cr")
            # step outputs (re-used every step; the returned tensors are fresh clones only where the
            # reference returns fresh tensors to the caller)
            f32 = dict(dtype=th.float32, device=self.device)
            self._ep_return = th.zeros(N, **f32)
            self._ep_length = th.zeros(N, dtype=th.int32, device=self.device)
            self._ep_flags = th.zeros(N, dtype=th.uint8, device=self.device)
            self._ep_past_gates = th.zeros(N, dtype=th.int32, device=self.device) if self.KIND == RACING else None
            self._terminal_gate = th.zeros(N, dtype=th.int32, device=self.device) if self.KIND == RACING else None
            self._terminal_obs = th.zeros((N, 13), **f32)
            self._gate = th.zeros(N, dtype=th.int32, device=self.device) if self.KIND == RACING else None

        state_size = 3 + 3 + 3 + (3 if dyn.angular_output_type == "euler" else 4)
        self.observation_space = spaces.Dict({"state": spaces.Box(low=-np.inf, high=np.inf, shape=(state_size,),
                                                                  dtype=np.float32)})
        self.action_space = spaces.Box(low=-1, high=1, shape=(4,), dtype=np.float32)
        self.render_mode = ["None" for _ in range(N)]
        self.deter = self.stoch = None
        self._is_initial = False
        self._info = [{"TimeLimit.truncated": False} for _ in range(N)]
        self._observations = TensorDict({})
        self._reward = th.zeros(N, device=self.device)
        self._done = th.zeros(N, dtype=th.bool, device=self.device)
        self._action = th.zeros((N, 4), device=self.device)
        self._qcache = self._ext_col = None
        self._half_step = False
        self._tape = None
        if requires_grad:
            self.set_requires_grad(True)
        self._outs = self._out(self._terminal_obs, self._ep_return, self._ep_flags)  # obs/reward/done patched per step
        self._outs_ref = C.byref(self._outs)
        self._vf_env_step = _lib.lib().vf_env_step
        self._ring, self._ring_i = [], 0
        if out_buffers:
            if out_buffers < 2:
                raise ValueError("out_buffers must be 0 (fresh tensors) or >= 2")
            with th.cuda.device(self.device):
                for _ in range(int(out_buffers)):
                    self._ring.append(_OutSlot(self))
        self._rollouts = {}      # step_n: K -> cached output buffers / launch descriptor / graph
        self._imu_cache = None

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return _lib.current_stream(self.device)

    def _out(self, obs, reward, done):
        o = _lib.EnvOut()
        o.obs, o.reward, o.done = _lib.ptr(obs), _lib.ptr(reward), _lib.ptr(done)
        o.ep_return, o.ep_length = _lib.ptr(self._ep_return), _lib.ptr(self._ep_length)
        o.ep_flags, o.terminal_obs = _lib.ptr(self._ep_flags), _lib.ptr(self._terminal_obs)
        o.gate = _lib.ptr(self._gate)
        o.ep_past_gates = _lib.ptr(self._ep_past_gates)
        o.terminal_gate = _lib.ptr(self._terminal_gate)
        dl = getattr(self, "_done_list", None)
        if dl is not None:
            o.done_list, o.done_count = dl[0].data_ptr(), dl[1].data_ptr()
        return o

    def enable_done_list(self, on: bool = True):
        """every step() additionally leaves the compacted list of the agents it finished (vf_env_out.done_list / done_count):
        ``done_indices()`` -> int32 tensor (unordered), what ``th.where(done)[0]`` costs a full pass over `done` for"""
        if on:
            self._done_list = (th.zeros(self.num_agent, dtype=th.int32, device=self.device), th.zeros(1, dtype=th.int32, device=self.device))
        else:
            self._done_list = None
        self._outs = self._out(self._terminal_obs, self._ep_return, self._ep_flags)
        self._outs_ref = C.byref(self._outs)
        if self._ring:
            with th.cuda.device(self.device):
                self._ring = [_OutSlot(self) for _ in self._ring]
        return self

    def done_indices(self):
        """indices of the agents the LAST step() finished (needs enable_done_list(); one host sync for the count)"""
        if getattr(self, "_done_list", None) is None:
            raise VisflyError("done_indices(): call enable_done_list() first")
        lst, cnt = self._done_list
        return lst[:int(cnt.item())]

    def _query(self):
        if self._qcache is None:
            N, dev = self.num_agent, self.device
            q = dict(step_count=th.empty(N, dtype=th.int32, device=dev), rewards=th.empty(N, device=dev),
                     flags=th.empty(N, dtype=th.uint8, device=dev), col_point=th.empty((N, 3), device=dev),
                     col_vec=th.empty((N, 3), device=dev), col_dis=th.empty(N, device=dev),
                     gate=th.empty(N, dtype=th.int32, device=dev), past_gates=th.empty(N, dtype=th.int32, device=dev))
            v = _lib.EnvView()
            for k, t in q.items():
                setattr(v, k, _lib.ptr(t))
            with th.cuda.device(dev):
                _lib.check(_lib.lib().vf_env_query(self._h, C.byref(v), self._stream()))
            ext = getattr(self, "_ext_col", None)
            if ext is not None:      # after step_finish(collision_point=...): the scene's closest point, except where the agent re-spawned
                cp, done = ext
                keep = done.view(-1, 1)
                q["col_point"] = th.where(keep, q["col_point"], cp)
                q["col_vec"] = th.where(keep, q["col_vec"], cp - self.envs.dynamics.position)
                q["col_dis"] = th.where(done, q["col_dis"], q["col_vec"].norm(dim=1))
            self._qcache = q
        return self._qcache

    def _static_obs(self, i=None):
        """observation entries that do not come out of the kernel (targets, gates)"""
        return {}

    def _terminal_static_obs(self, i):
        """the same entries as they were in the terminal (pre-reset) observation of agent i"""
        return self._static_obs(i)

    def _terminal_state_obs(self, tobs, i):
        """ "state" entry of agent i's terminal observation from the (N,13) rows the step kernel wrote where done (already in
        the env's obs_mode); envs that assemble their observation on the host override this"""
        return tobs[i]

    @property
    def state_slab(self):
        """the slab without the prefetched re-spawn copies (which launch path refilled them last is not part of the env's state)"""
        return self._slab[:, :self._slab.shape[1] - (8 if self._ecfg.spawn_prefetch else 0)]

    def _terminal_state_rows(self):
        """(N, w) "state" rows of the terminal observations in the env's own observation map, valid where `done` was set by the
        last step (what a trainer values for the TimeLimit bootstrap); the step kernel writes them in obs_mode already"""
        return self._terminal_obs

    def _state_obs(self, raw_state):
        """observation "state" from the raw (N,13) dynamics state; the step kernel applies the same map on the
        device (vf_env_cfg.obs_mode), this host version only runs after resets"""
        return raw_state

    def _full_obs(self, state, raw=False):
        obs = TensorDict({"state": self._state_obs(state) if raw else state})
        obs.update(self._static_obs())
        return obs

    # ------------------------------------------------------------------ reset
    def _reset_kernel(self, idx, fs):
        with th.cuda.device(self.device):
            di = None if idx is None else th.as_tensor(idx, dtype=th.int32).reshape(-1).to(self.device).contiguous()
            k = self.num_agent if di is None else di.numel()
            dfs = None
            if fs is not None:
                dfs = th.as_tensor(fs, dtype=th.float32).to(self.device).reshape(k, 22).contiguous()
            _lib.check(_lib.lib().vf_env_reset(self._h, _lib.ptr(di), k, _lib.ptr(dfs), self._stream()))
            self._keep = (di, dfs)
        self._qcache = self._ext_col = None

    def _replay_states(self, k, indexed):
        """host replay of the reference's draw order for k agents -> (k,22) full states
        (droneEnv.py:237-251, dynamics.py:229-256)"""
        dyn = self.envs.dynamics
        p, q, v, w = self._spawner.generate(k)
        fs = th.zeros((k, 22))
        fs[:, 0:3], fs[:, 3:7], fs[:, 7:10], fs[:, 10:13] = p, q, v, w
        fs[:, 13:17] = float(dyn.constants["w_init"])
        fs[:, 17:21] = float(dyn.constants["T_init"])
        if indexed:
            fs[:, 21] = th.zeros((k,)) + th.rand((k,), generator=dyn.rng) * 3.14 * 2    # dynamics.py:256
        return fs

    def _consume_imu_noise(self):
        """the reference draws th.rand(N,13) for the IMU noise at every update_observation (droneEnv.py:114-116,333;
        utils/type.py:37-38) -- also when its amplitude is zero; replay mode keeps the shared stream aligned and keeps
        the draw for sensor_obs["IMU"]"""
        u = th.rand((self.num_agent, 13), generator=self.envs.dynamics.rng)
        self._imu_cache = None
        self._imu_draw = u if self._imu_noise is not None else None

    def _imu_obs(self):
        """sensor_obs["IMU"] = state + noise, quaternion re-normalised (droneEnv.py:114-125).  Nothing on the visual=False
        observation path reads it (HoverEnv.py:62-70), so it is formed on demand, once per step: replay mode uses the
        draw the reference would have made at this step, device mode draws from the device generator."""
        st = self.envs.dynamics.state
        if self._imu_noise is None:
            return st
        if self._imu_cache is None:
            mean, half = (x.to(self.device) for x in self._imu_noise)
            if self.spawn_mode == "replay" and getattr(self, "_imu_draw", None) is not None:
                u = self._imu_draw.to(self.device)
            else:
                u = th.rand((self.num_agent, 13), device=self.device)
            noisy = st + ((u - 0.5) * half + mean)                                          # utils/type.py:37-38
            noisy = th.cat([noisy[:, :3], th.nn.functional.normalize(noisy[:, 3:7], p=2, dim=1), noisy[:, 7:]], dim=1)
            self._imu_cache = noisy
        return self._imu_cache

    def reset(self, state=None, **_unused):
        """DroneGymEnvsBase.reset (droneGymEnv.py:302-327) -> observations"""
        self._is_initial = True
        self._half_step = False
        if state is not None:
            fs = th.as_tensor(state, dtype=th.float32)
            if fs.shape != (self.num_agent, 22):
                raise ValueError("reset(state=...) expects the (N,22) full_state layout")
            self._reset_kernel(None, fs)
            if self.spawn_mode == "replay":
                self._consume_imu_noise()
        elif self.spawn_mode == "replay":
            self._reset_kernel(None, self._replay_states(self.num_agent, indexed=False))
            if self._drag_random:   # Dynamics.reset draws ONE (3,1) factor pair for all agents (dynamics.py:244-246)
                dyn, r = self.envs.dynamics, self._drag_random
                fl = ((th.rand((3, 1), generator=dyn.rng) - 0.5) * 2 * r).clamp(-0.5, .5) + 1
                fq = ((th.rand((3, 1), generator=dyn.rng) - 0.5) * 2 * r).clamp(-0.5, .5) + 1
                kl, kq = dyn._drag_rows(fl.reshape(1, 3), fq.reshape(1, 3), 1)
                gd = dyn._G - 2 - (1 if self.KIND == RACING else 0)
                self._slab[:, gd, :, 1:4] = kl.to(self.device)
                self._slab[:, gd + 1, :, 1:4] = kq.to(self.device)
            self._consume_imu_noise()
        else:
            self._reset_kernel(None, None)
        self._info = [{"TimeLimit.truncated": False, "episode_done": False} for _ in range(self.num_agent)]
        self._reward = th.zeros(self.num_agent, device=self.device)
        self._done = th.zeros(self.num_agent, dtype=th.bool, device=self.device)
        self._observations = self._full_obs(self.envs.dynamics.state, raw=True)
        return self._format_obs(self._observations)

    def reset_agent_by_id(self, agent_indices=None, state=None, reset_obs=None):
        """droneGymEnv.py:339-349: re-spawn the given agents, clear their counters"""
        assert not isinstance(agent_indices, bool)
        idx = th.arange(self.num_agent) if agent_indices is None else th.as_tensor(agent_indices).reshape(-1).cpu()
        if state is not None:
            fs = th.as_tensor(state, dtype=th.float32)
        elif self.spawn_mode == "replay":
            fs = self._replay_states(len(idx), indexed=True)
        else:
            fs = None
        self._reset_kernel(idx, fs)
        if self.spawn_mode == "replay":
            self._consume_imu_noise()
        self._observations = self._full_obs(self.envs.dynamics.state, raw=True)
        for i in idx.tolist():
            self._info[i] = {"TimeLimit.truncated": False, "episode_done": False}
        return self._observations

    def reset_env_by_id(self, scene_indices=None):
        """droneGymEnv.py:329-337: re-spawn every agent of the given scenes"""
        assert not isinstance(scene_indices, bool)
        sc = th.arange(self.num_scene) if scene_indices is None else th.atleast_1d(th.as_tensor(scene_indices)).cpu()
        agents = (th.arange(self.num_agent_per_scene).unsqueeze(0) + sc.unsqueeze(1) * self.num_agent_per_scene).flatten()
        return self.reset_agent_by_id(agents)

    def examine(self):
        if bool(self._done.any()):
            self.reset_agent_by_id(th.where(self._done)[0])
        return self._observations

    # ------------------------------------------------------------------ step
    def step(self, _action, is_test=False, **_unused):
        """DroneGymEnvsBase.step (droneGymEnv.py:141-218) -> (obs, reward, done, info).  With
        requires_grad=True the returned state observation and reward are attached to the autograd graph
        (droneGymEnv.py:209-213); their backward is the adjoint kernel (visfly_amd/bptt.py)."""
        if self.requires_grad and isinstance(_action, th.Tensor) and th.is_grad_enabled():
            from ..bptt import EnvStepFunction
            if not self.tensor_output:
                raise ValueError("requires_grad should be False if tensor_output is False")          # :211-212
            state, reward, self._token = EnvStepFunction.apply(_action, self._token, self, is_test)
            obs0, done, info = self._last_step_aux
            obs = self._full_obs(state)
            self._observations = obs
            return obs, reward, done, info
        return self._step_no_grad(_action, is_test)

    def _step_no_grad(self, _action, is_test=False, record=False, borrow=False, prefilled=False):
        # borrow (trainers that own their per-horizon buffers): the tape keeps a REFERENCE to the action tensor and the step's
        # done flags are written straight into the tape row (returned as such) -- two device copies per step less;
        # prefilled: tape row `_tape_t` already holds the current slab (vf_bptt_accumulate_checkpoint after the previous step)
        assert self._is_initial, "You should call reset() before step()"
        if self._half_step:
            raise VisflyError("step(): step_begin() is waiting for its step_finish()")
        if self.envs.dynamics._wind_fn is not None:
            if self._tape is not None:
                raise NotImplementedError("string wind functions have no adjoint (the tape does not record the wind rows)")
            self.envs.dynamics.update_wind()                                                # dynamics.py:320
        N, dev = self.num_agent, self.device
        a = _action
        if not (isinstance(a, th.Tensor) and a.is_cuda and a.dtype == th.float32 and a.dim() == 2
                and a.shape[0] == N and a.is_contiguous()):
            a = a if isinstance(a, th.Tensor) else th.as_tensor(np.asarray(a))
            a = a.to(dev, dtype=th.float32).reshape(N, 4).contiguous()
        if self.validate_actions:
            assert a.max() <= 1 and a.min() >= -1                                           # :144
        self._action = a
        if _cuda_get_device() != dev.index:
            th.cuda.set_device(dev)
        replay = self.spawn_mode == "replay"
        if self._ring and self._tape is None and not replay and self.tensor_output:
            # zero-allocation path: pre-built output set, one ctypes call, no per-step Python objects
            slot = self._ring[self._ring_i]
            self._ring_i = self._ring_i + 1 if self._ring_i + 1 < len(self._ring) else 0
            rc = self._vf_env_step(self._h, a.data_ptr(), slot.ref, 0 if is_test else 1, _raw_stream(dev.index))
            if rc:
                _lib.check(rc)
            self._qcache = self._imu_cache = self._ext_col = None
            obs = slot.obs
            if obs is None or not self._STATIC_OBS_CONST:
                obs = slot.obs = self._full_obs(slot.state)
            info = slot.info
            if info._ready:
                info._clear()
            info._src = (self, slot.done, self._ep_return, self._ep_length, self._ep_flags, self._terminal_obs,
                         self._extra_info())
            self._reward, self._done, self._observations = slot.reward, slot.done, obs
            return obs, slot.reward, slot.done, info
        state = th.empty((N, 13), dtype=th.float32, device=dev)
        reward = th.empty(N, dtype=th.float32, device=dev)
        tape_t = -1
        if self._tape is not None and (record or self._record_all):   # checkpoint for the adjoint pass
            tape_t = self._tape_t
            if tape_t >= self._tape.shape[0]:
                raise VisflyError("tape is full: call env.detach() (BPTT horizon exceeded)")
            if not prefilled:          # prefilled: the caller's previous launch already checkpointed the slab into this row
                self._tape[tape_t].copy_(self._slab)
            if borrow:
                self._tape_action_ref[tape_t] = a
            else:
                self._tape_action_ref[tape_t] = None
                self._tape_actions[tape_t].copy_(a)
            self._tape_t += 1
            self._substep_range = None        # a step recorded launch by launch: the sub-step tape no longer covers the horizon
        borrow_done = borrow and tape_t >= 0
        done = self._tape_done[tape_t] if borrow_done else th.empty(N, dtype=th.bool, device=dev)      # the kernel writes 0/1 bytes
        o = self._outs
        o.obs, o.reward, o.done = state.data_ptr(), reward.data_ptr(), done.data_ptr()
        rc = self._vf_env_step(self._h, a.data_ptr(), self._outs_ref, 0 if (is_test or replay) else 1,
                               _raw_stream(dev.index))
        if rc:
            _lib.check(rc)
        self._qcache = self._imu_cache = self._ext_col = None
        if tape_t >= 0 and not borrow_done:
            self._tape_done[tape_t].copy_(done)
        self._reward, self._done = reward, done
        self._observations = obs = self._full_obs(state)
        info = _Info(self, done, self._ep_return, self._ep_length, self._ep_flags, self._terminal_obs,
                     self._extra_info())
        if replay:
            self._consume_imu_noise()
            if not is_test:
                info._build()  # the info dicts of done agents are collected before the reset (:197-208)
                idx = th.where(done)[0]
                if idx.numel():
                    self.reset_agent_by_id(idx)
                obs = self._observations
        if self.tensor_output:
            return obs, reward, done, info
        return self._format_obs(obs), reward.cpu().numpy(), done.cpu().numpy().astype(np.int32), info   # :218

    # ------------------------------------------------------------------ multi-step launch (open-loop action sequences)
    def step_n(self, actions, is_test=False, graph=False, fused=False):
        """K consecutive step() calls with the launch loop in C (vf_env_step_n): `actions` is a (K,N,4) device tensor.
        Returns (obs (K,N,13), reward (K,N), done (K,N)) -- row k is what the k-th step()
        would have returned (obs after auto-reset, reward / done before); bit-identical to K step() calls.  No info dicts: the
        episode outputs (ep_return / ep_length / ep_flags / terminal_obs) are ONE buffer per env, so an agent that finishes twice
        within the K steps keeps only its last episode's entries -- trainers that need every episode's info use step().  The output
        buffers are cached per K and re-used by the next step_n call of the same K.  graph=True replays the K launches from
        a hipGraph captured on the first call for this (K, actions buffer): the caller refills that SAME actions tensor
        between calls.  fused=True runs the K steps inside ONE launch with the agents held in registers between the steps
        (vf_env_rollout_fused: no per-step launch boundary, no per-step state round trip; same results).  The reference has
        no counterpart (its loop is `for a in seq: env.step(a)`, e.g.
        utils/evaluate.py:62-103); not available in replay-spawn mode or while a BPTT tape is recording."""
        assert self._is_initial, "You should call reset() before step()"
        if self.spawn_mode == "replay":
            raise VisflyError("step_n: spawn='replay' needs a host round trip per step; use step()")
        if self._tape is not None:
            raise VisflyError("step_n: a BPTT tape is recording (requires_grad / enable_tape); use step()")
        if getattr(self, "_HOST_OBS", False):
            raise VisflyError(f"step_n: {type(self).__name__} assembles its observation on the host per step; use step()")
        if self.envs.dynamics._wind_fn is not None:
            raise VisflyError("step_n: string wind functions are re-evaluated on the host before every step; use step()")
        N, dev = self.num_agent, self.device
        a = actions
        if not (isinstance(a, th.Tensor) and a.is_cuda and a.dtype == th.float32 and a.is_contiguous()):
            a = th.as_tensor(np.asarray(a) if not isinstance(a, th.Tensor) else a).to(dev, dtype=th.float32).contiguous()
        if a.dim() != 3 or a.shape[1:] != (N, 4):
            raise ValueError(f"step_n expects actions of shape (K,{N},4), got {tuple(a.shape)}")
        K = a.shape[0]
        if self.validate_actions:
            assert a.max() <= 1 and a.min() >= -1
        if _cuda_get_device() != dev.index:
            th.cuda.set_device(dev)
        ro = self._rollouts.get(K)
        if ro is None:
            while len(self._rollouts) >= 4:                      # bounded cache: output buffers are K x N x 18 floats each
                old = self._rollouts.pop(next(iter(self._rollouts)))
                for gph in old["graphs"].values():
                    _lib.lib().vf_env_graph_destroy(gph)
            obs = th.empty((K, N, 13), dtype=th.float32, device=dev)
            reward = th.empty((K, N), dtype=th.float32, device=dev)
            done = th.empty((K, N), dtype=th.bool, device=dev)
            r = _lib.EnvRollout()
            r.out = self._out(obs, reward, done)
            r.action_stride, r.obs_stride, r.reward_stride, r.done_stride = 4 * N, 13 * N, N, N
            r.K = K
            ro = self._rollouts[K] = {"obs": obs, "reward": reward, "done": done, "r": r, "ref": C.byref(r), "graphs": {}}
        r = ro["r"]
        r.actions, r.auto_reset = a.data_ptr(), 0 if is_test else 1
        L = _lib.lib()
        if graph:
            key = (a.data_ptr(), r.auto_reset, int(L.vf_env_ring_phase(self._h)))   # one graph per delay-ring phase
            g = ro["graphs"].get(key)
            if g is None:
                g = _lib._vp()
                _lib.check(L.vf_env_graph_create(self._h, ro["ref"], C.byref(g)))
                ro["graphs"][key] = g
                ro.setdefault("keep", []).append(a)          # the graph holds this buffer's address
            rc = L.vf_env_graph_launch(g, _raw_stream(dev.index))
        elif fused:
            rc = L.vf_env_rollout_fused(self._h, ro["ref"], _raw_stream(dev.index))
        else:
            rc = L.vf_env_step_n(self._h, ro["ref"], _raw_stream(dev.index))
        if rc:
            _lib.check(rc)
        self._qcache = self._imu_cache = self._ext_col = None
        self._action = a[K - 1]
        self._reward, self._done = ro["reward"][K - 1], ro["done"][K - 1]
        self._observations = self._full_obs(ro["obs"][K - 1])
        return ro["obs"], ro["reward"], ro["done"]

    # ------------------------------------------------------------------ the step split around an external scene manager
    def step_begin(self, _action, pose_out=None):
        """first half of DroneEnvsBase.step with visual=True (droneEnv.py:374-377): the dynamics interval alone, then the pose an
        external renderer / scene manager takes through set_pose -> the dict of export_pose().  Follow with step_finish().
        Not available in replay-spawn mode, while a BPTT tape is recording, or with IMU noise switched on."""
        assert self._is_initial, "You should call reset() before step()"
        if self.spawn_mode == "replay" or self._tape is not None:
            raise VisflyError("step_begin: needs spawn='device' and no recording tape")
        N, dev = self.num_agent, self.device
        a = _action
        if not (isinstance(a, th.Tensor) and a.is_cuda and a.dtype == th.float32 and a.dim() == 2
                and a.shape[0] == N and a.is_contiguous()):
            a = a if isinstance(a, th.Tensor) else th.as_tensor(np.asarray(a))
            a = a.to(dev, dtype=th.float32).reshape(N, 4).contiguous()
        if self.validate_actions:
            assert a.max() <= 1 and a.min() >= -1                                           # droneGymEnv.py:144
        self._action = a
        self.envs.dynamics.step(a)                                                          # droneEnv.py:375
        self._qcache = self._imu_cache = self._ext_col = None
        if hasattr(self, "_g_obs"):
            self._g_obs = None              # RacingEnv: the split step reports the env's current gate index
        self._half_step = True
        return self.export_pose(pose_out)

    def step_finish(self, collision_point=None, is_out_bounds=None, is_test=False):
        """second half: update_collision with the scene manager's answer for the poses step_begin returned -- `collision_point`
        (N,3) = its closest scene point per agent, `is_out_bounds` (N,) bool (droneEnv.py:330-342; either None = the bounding
        box of this library) -- then everything DroneGymEnvsBase.step does after the simulator step (droneGymEnv.py:161-218).
        Returns (obs, reward, done, info) like step().  Re-spawned agents take the bounding-box query until the next step."""
        if not getattr(self, "_half_step", False):
            raise VisflyError("step_finish: call step_begin(action) first")
        N, dev = self.num_agent, self.device
        cp = ob = None
        if collision_point is not None:
            cp = th.as_tensor(collision_point).to(dev, dtype=th.float32).reshape(N, 3).contiguous()
        if is_out_bounds is not None:
            ob = th.as_tensor(is_out_bounds).to(dev).reshape(N).to(th.uint8).contiguous()
        state = th.empty((N, 13), dtype=th.float32, device=dev)
        reward = th.empty(N, dtype=th.float32, device=dev)
        done = th.empty(N, dtype=th.bool, device=dev)
        o = self._outs
        o.obs, o.reward, o.done = state.data_ptr(), reward.data_ptr(), done.data_ptr()
        with th.cuda.device(dev):
            _lib.check(_lib.lib().vf_env_finish_step(self._h, _lib.ptr(cp), _lib.ptr(ob), self._outs_ref, 0 if is_test else 1,
                                                     self._stream()))
        self._half_step = False
        self._qcache = self._imu_cache = None
        self._ext_col = (cp, done) if cp is not None else None     # collision_point / _vector / _dis properties follow the scene
        self._reward, self._done = reward, done
        self._observations = obs = self._full_obs(state)
        info = _Info(self, done, self._ep_return, self._ep_length, self._ep_flags, self._terminal_obs, self._extra_info())
        if self.tensor_output:
            return obs, reward, done, info
        return self._format_obs(obs), reward.cpu().numpy(), done.cpu().numpy().astype(np.int32), info

    def export_pose(self, out=None):
        """what DroneEnvsBase.step hands to sceneManager.set_pose when visual=True (droneEnv.py:375-377): AoS device tensors
        {"position" (N,3), "rotation" (N,4) wxyz, "velocity" (N,3) incl. wind, "angular_velocity" (N,3)} of the CURRENT state,
        written by one launch (vf_env_export_pose).  `out`: a dict from a previous call to refill in place."""
        N, dev = self.num_agent, self.device
        if out is None:
            out = {"position": th.empty((N, 3), device=dev), "rotation": th.empty((N, 4), device=dev),
                   "velocity": th.empty((N, 3), device=dev), "angular_velocity": th.empty((N, 3), device=dev)}
        with th.cuda.device(dev):
            _lib.check(_lib.lib().vf_env_export_pose(self._h, _lib.ptr(out["position"]), _lib.ptr(out["rotation"]),
                                                     _lib.ptr(out["velocity"]), _lib.ptr(out["angular_velocity"]),
                                                     self._stream()))
        return out

    def _extra_info(self):
        return None

    def _format_obs(self, obs):
        if self.tensor_output:
            return obs
        return TensorDict({k: v.detach().cpu().numpy() for k, v in obs.items()})

    # ------------------------------------------------------------------ adjoint (first-order / BPTT) support
    def enable_tape(self, horizon: int):
        """record what the reverse pass needs: per step a copy of the slab (pre-step state), the action
        and the done flags.  Replaces the autograd graph requires_grad=True builds in the reference."""
        dev = self.device
        self._tape = th.empty((horizon,) + tuple(self._slab.shape), dtype=th.float32, device=dev)
        self._tape_actions = th.empty((horizon, self.num_agent, 4), dtype=th.float32, device=dev)
        self._tape_done = th.zeros((horizon, self.num_agent), dtype=th.bool, device=dev)
        self._tape_action_ref = [None] * horizon          # borrowed action tensors (see _step_no_grad)
        self._adj = th.zeros_like(self._slab)
        self._tape_t = 0
        self._record_all = True      # manual mode: every step() is recorded until clear_tape()/detach()
        # the sub-step tape is sized by the tape: a new horizon gets a new one (a second trainer with a longer horizon on
        # the same env handed vf_bptt_rollout the first trainer's rows)
        self._substep, self._substep_range = None, None

    def rollout_policy(self, policy, obs_keys, eps, actions, d_reward, loss, disc, gamma, scale, reward_rows=None, ep_flag_rows=None):
        """H = eps.shape[0] closed-loop control steps -- policy forward (slots 0..H-1 of `policy`, reserved back to back), action
        head, state checkpoint on the tape, fused env step, loss / discount recurrence -- in ONE persistent launch
        (vf_bptt_rollout).  Leaves exactly what H rounds of policy.forward_act + _step_no_grad(record=True) +
        vf_bptt_accumulate leave.  A policy with two 4-wide heads and no log_std parameter is the reference's own Actor
        (td_policies.py:146-252): state-dependent log_std, what H rounds of policy.forward + vf_shac_head_fwd leave, the heads of
        every step in the slots' "mean" / "value" buffers.  -> False when the library has no roll-out kernel for this env / network
        / dynamics configuration (the caller then steps launch by launch).  ``reward_rows`` (H,N) / ``ep_flag_rows`` (H,N) uint8:
        optional per-step copies of the reward and of the episode flags (valid where done) -- SHAC's horizon buffer."""
        W = self._OBS_W        # 13, or the 16 gate-relative columns the persistent launches form themselves for RacingEnv2 (VF_OBS_RACE2)
        if (self._tape is None or self.spawn_mode != "device" or self._imu_noise is not None or self._half_step
                or self.envs.dynamics._wind_fn is not None or (getattr(self, "_HOST_OBS", False) and W == 13) or not self.tensor_output
                or getattr(self, "obs_gate_exact", False)):
            return False
        H, N, dev = eps.shape[0], self.num_agent, self.device
        t0 = self._tape_t
        if t0 + H > self._tape.shape[0]:
            raise VisflyError("tape is full: call env.detach() (BPTT horizon exceeded)")
        nblk, blk = policy._slot_blocks.get(N, (0, None))
        if nblk < H or any(k not in ("state", "target") for k in obs_keys):
            return False
        obs = self.get_observation()
        blk["obs:state"][0].copy_(obs["state"].detach())
        o1 = None
        if "target" in obs_keys:
            blk["obs:target"][:H].copy_(obs["target"].detach().unsqueeze(0).expand(H, N, -1))     # constant per env
            o1 = blk["obs:target"]
        b0 = policy._buffers(N, 0)
        key = (N, 0, True)
        d = policy._descs.get(key)
        if d is None:
            d = policy._descs[key] = policy._fused_desc(b0, True)
        policy._pack()
        if getattr(self, "_roll_out", None) is None:
            self._roll_scratch = (th.empty((N, W), dtype=th.float32, device=dev), th.empty(N, dtype=th.float32, device=dev),
                                  th.empty(N, dtype=th.bool, device=dev))
            self._roll_out = self._out(*self._roll_scratch)
            if W != 13:         # the launch's terminal rows are W wide as well: a buffer of their own
                self._terminal_rows_w = th.zeros((N, W), dtype=th.float32, device=dev)
                self._roll_out.terminal_obs = _lib.ptr(self._terminal_rows_w)
        final = th.empty((N, W), dtype=th.float32, device=dev)
        L = _lib.lib()
        # sub-step tape (include/visfly_amd.h, vf_bptt_rollout): what the reverse launch reads instead of replaying every interval;
        # one (S + 3 rows, waves of 16 agents, 64) float4 record block per tape row, allocated with the first persistent roll-out
        # ... and only where the reverse launch reads it (vf_bptt_reverse: 16 agents per wave, i.e. N <= 16 384 per launch, and a
        # delay ring of at most 4 slots): elsewhere it would be (S + 3) KiB per wave-step of stores nobody loads
        sub = None
        if self.substep_tape and N <= 16384 and self.envs.dynamics._comm_delay_steps <= 4:
            S = int(self.envs.dynamics.constants["interval_steps"])
            shape = (self._tape.shape[0], S + 3, (N + 15) // 16, 64, 4)
            if getattr(self, "_substep", None) is None or tuple(self._substep.shape) != shape:
                self._substep = th.empty(shape, dtype=th.float32, device=dev)
            sub = self._substep[t0]
        self._substep_range = None
        two_heads = tuple(policy.head_dims) == (4, 4) and policy.log_std.numel() == 0
        self._ensure_bptt_plugin(policy)
        with th.cuda.device(dev):
            rc = L.vf_bptt_rollout(self._h, C.byref(d), _lib.ptr(policy.flat), _lib.ptr(policy._packed), _lib.ptr(blk["obs:state"]),
                                   _lib.ptr(o1), None if two_heads else _lib.ptr(policy.log_std), _lib.ptr(eps), _lib.ptr(actions),
                                   C.byref(self._roll_out), _lib.ptr(final), _lib.ptr(self._tape[t0]), self._slab.numel(),
                                   self._tape_done[t0].data_ptr(), _lib.ptr(d_reward), _lib.ptr(loss), _lib.ptr(disc), float(gamma),
                                   float(scale), H, _lib.ptr(sub), _lib.ptr(blk["mean"]) if two_heads else None,
                                   _lib.ptr(blk["value"]) if two_heads else None, _lib.ptr(reward_rows),
                                   None if ep_flag_rows is None else ep_flag_rows.data_ptr(), self._stream())
        if rc == _lib.EUNSUPPORTED:
            _lib.warn_unsupported("vf_bptt_rollout")
            return False
        if rc:
            _lib.check(rc)
        if sub is not None:
            self._substep_range = (t0, t0 + H)
        for t in range(H):
            self._tape_action_ref[t0 + t] = actions[t]
        self._tape_t = t0 + H
        self._qcache = self._imu_cache = self._ext_col = None
        self._action = actions[H - 1]
        self._reward, self._done = self._roll_scratch[1], self._tape_done[t0 + H - 1]
        self._after_persistent_launch()
        self._observations = self._full_obs(final)
        policy._last_M, policy._last_slot = N, H - 1
        return True

    def _ensure_bptt_plugin(self, policy):
        """a generated actor class: the two persistent launches of its horizons are one more plugin, per env kind / action type /
        integrator / motor lag (visfly_amd/_jit.py: ensure_bptt; ~2 min of hipcc on first use, 16 agents per wave: N <= 16 384)"""
        if not getattr(policy, "chain_jit", False) or self.num_agent > 16384:
            return
        try:
            from .. import _jit
            c = self.envs.dynamics.constants
            kind = 3 if self._OBS_W == 16 else self.KIND       # kernel-side kind: VF_ENV_RACING2 forms RacingEnv2's 16 columns itself
            _jit.ensure_bptt(policy.chain_shape, (kind, int(c["action_type"]), int(c["integrator"]), bool(c["ctrl_delay"])))
        except Exception as e:          # no hipcc, ...: launch by launch, with the library's warning
            import warnings
            warnings.warn(f"visfly_amd: no BPTT plugin for this network ({e})")

    def _after_persistent_launch(self):
        """host-side caches of per-step state that a persistent launch stepped past (overridden where an env keeps any)"""

    def collect_policy(self, policy, obs_keys, buf, boot, noise_key, sample_step, last_starts):
        """T = buf.actions.shape[0] rounds of PPO's collect_rollouts loop -- policy.forward (both heads), squashed-Gaussian
        sample + log-prob (Philox counters sample_step + 1 ..), env.step, RolloutBuffer rows, TimeLimit list / episode
        statistics -- in ONE persistent launch (vf_ppo_rollout).  buf.obs["state"][0] (and every row of buf.obs["target"])
        and buf.episode_starts[0] are the caller's.  -> the final observation TensorDict, or False when the library has no
        roll-out kernel for this env / network / dynamics configuration (the caller then steps launch by launch)."""
        W = self._OBS_W        # 13, or RacingEnv2's 16 columns (formed inside the launch: VF_OBS_RACE2)
        if (self._tape is not None or self.spawn_mode != "device" or self._imu_noise is not None or self._half_step
                or self.envs.dynamics._wind_fn is not None or (getattr(self, "_HOST_OBS", False) and W == 13) or not self.tensor_output
                or getattr(self, "obs_gate_exact", False)
                or (W == 13 and self._terminal_state_rows() is not self._terminal_obs)
                or any(k not in ("state", "target") for k in obs_keys) or policy._plan is None or not policy.fused
                or policy.obs_dims.get("state") != W or buf.obs["state"].shape[-1] != W):
            return False
        assert self._is_initial, "You should call reset() before step()"
        T, N, dev = buf.actions.shape[0], self.num_agent, self.device
        b0 = policy._buffers(N, 0)
        key = (N, 0, False)
        d = policy._descs.get(key)
        if d is None:
            d = policy._descs[key] = policy._fused_desc(b0, False)
        policy._pack()
        sc = getattr(self, "_collect_scratch", None)
        if sc is None:
            sc = self._collect_scratch = {"reward": th.empty(N, dtype=th.float32, device=dev),
                                          "done": th.empty(N, dtype=th.bool, device=dev),
                                          "state": th.empty((N, W), dtype=th.float32, device=dev)}
            sc["out"] = self._out(sc["state"], sc["reward"], sc["done"])
            if W != 13:         # the launch's terminal rows are W wide as well: a buffer of their own
                sc["terminal"] = th.zeros((N, W), dtype=th.float32, device=dev)
                sc["out"].terminal_obs = _lib.ptr(sc["terminal"])
        final = th.empty((N, W), dtype=th.float32, device=dev)
        o1 = buf.obs["target"] if "target" in obs_keys else None
        a = _lib.PpoRolloutArgs()
        a.T, a.w1, a.capacity = T, 0 if o1 is None else o1.shape[-1], boot["cap"]
        a.obs_state, a.obs_target = _lib.ptr(buf.obs["state"]), _lib.ptr(o1)
        a.obs_target_row = None if o1 is None else _lib.ptr(o1[0])
        a.obs_final, a.means, a.values = _lib.ptr(final), None, _lib.ptr(buf.values)
        a.actions, a.log_probs, a.rewards = _lib.ptr(buf.actions), _lib.ptr(buf.log_probs), _lib.ptr(buf.rewards)
        a.episode_starts, a.last_starts, a.log_std = _lib.ptr(buf.episode_starts), _lib.ptr(last_starts), _lib.ptr(policy.log_std)
        a.noise_key, a.sample_step = noise_key, sample_step
        a.cursor, a.idx_list, a.rows0 = boot["cursor"].data_ptr(), boot["idx"].data_ptr(), _lib.ptr(boot["rows0"])
        a.rows1, a.stat = _lib.ptr(boot["rows1"]), _lib.ptr(boot["stat"])
        a.out = C.addressof(sc["out"])
        if getattr(policy, "chain_jit", False):
            # a generated chain class: its roll-out launch is one more plugin, per env kind / action type / integrator / motor lag
            # (visfly_amd/_jit.py; ~15 s of hipcc on first use)
            try:
                from .. import _jit
                c = self.envs.dynamics.constants
                kind = 3 if W == 16 else self.KIND          # kernel-side kind: VF_ENV_RACING2 forms RacingEnv2's 16 columns itself
                _jit.ensure_rollout(policy.chain_shape, (kind, int(c["action_type"]), int(c["integrator"]), bool(c["ctrl_delay"])))
            except Exception as e:          # no hipcc, ...: launch by launch, with the warning below
                import warnings
                warnings.warn(f"visfly_amd: no roll-out plugin for this network ({e})")
        with th.cuda.device(dev):
            rc = _lib.lib().vf_ppo_rollout(self._h, C.byref(d), _lib.ptr(policy.flat), _lib.ptr(policy._packed), C.byref(a), self._stream())
        if rc == _lib.EUNSUPPORTED:
            _lib.warn_unsupported("vf_ppo_rollout")
            return False
        if rc:
            _lib.check(rc)
        self._qcache = self._imu_cache = self._ext_col = None
        self._action = buf.actions[T - 1]
        self._reward, self._done = sc["reward"], sc["done"]
        self._after_persistent_launch()
        self._observations = obs = self._full_obs(final)
        return obs

    def reverse_policy(self, policy, H, eps, actions, d_reward, d_means, g_log_std, d_log_stds=None):
        """the reverse half of the last H recorded steps in ONE persistent launch (vf_bptt_reverse): for t = H-1 .. 0 the adjoint of
        env step t and the policy's action-head reverse + reverse chain of slot t.  -> False when the library has no kernel for
        this configuration (the caller then sweeps launch by launch).  d_means (H,N,4) out, g_log_std (H,N,4) zeroed in / out.
        ``d_log_stds`` (H,N,4) out instead of g_log_std: the reference's Actor (two heads, see rollout_policy) -- both trunks are
        swept, the head gradients of every step land in d_means / d_log_stds."""
        N, dev = self.num_agent, self.device
        t0 = self._tape_t - H
        nblk, blk = policy._slot_blocks.get(N, (0, None))
        if self._tape is None or t0 < 0 or nblk < H:
            return False
        two_heads = d_log_stds is not None
        key = ("bwd_flat", N, H, two_heads)
        cached = policy._descs.get(key)
        if cached is None:
            b = {name: t[:H].reshape(-1, t.shape[-1]) for name, t in blk.items()}
            d, d_in = policy._bwd_desc(b, H * N, d_means.view(-1, 4), d_log_stds.view(-1, 4) if two_heads else None, True)
            cached = policy._descs[key] = (d, d_in, th.empty((H, N, 4), dtype=th.float32, device=dev))
        d, d_in, d_action = cached
        im, iv = policy._head_entries(two_heads)
        d.layer[im].dY = _lib.ptr(d_means)
        if two_heads:
            d.layer[iv].dY = _lib.ptr(d_log_stds)
        policy._pack()
        # the sub-step tape only if exactly these H steps were recorded by ONE persistent roll-out with the tape on
        sub = self._substep[t0] if getattr(self, "_substep_range", None) == (t0, t0 + H) else None
        with th.cuda.device(dev):
            rc = _lib.lib().vf_bptt_reverse(self._h, C.byref(d), _lib.ptr(policy._packed), None if two_heads else _lib.ptr(policy.log_std),
                                            _lib.ptr(eps), _lib.ptr(actions), _lib.ptr(self._tape[t0]), self._slab.numel(),
                                            self._tape_done[t0].data_ptr(), _lib.ptr(d_reward), _lib.ptr(self._adj), _lib.ptr(d_action),
                                            _lib.ptr(d_in["state"]), None if two_heads else _lib.ptr(g_log_std), H, _lib.ptr(sub),
                                            _lib.ptr(blk["value"]) if two_heads else None, self._stream())
        if rc == _lib.EUNSUPPORTED:
            _lib.warn_unsupported("vf_bptt_reverse")
            return False
        if rc:
            _lib.check(rc)
        return True

    def clear_tape(self):
        """env.detach() of the reference (droneGymEnv.py:286-300): cut the graph at the current state"""
        if self._tape is not None:
            self._tape_t = 0
            self._substep_range = None
            self._adj.zero_()
            self._token = th.zeros(1, device=self.device, requires_grad=True)

    def backward_step(self, t: int, d_obs=None, d_reward=None):
        """reverse pass of recorded step t; call for t = last .. 0.  d_obs (N,13) / d_reward (N,) are the
        loss gradients w.r.t. what step t returned; returns dLoss/d(action of step t) (N,4)."""
        if self._tape is None or not (0 <= t < self._tape_t):
            raise VisflyError("backward_step: no recorded step %d" % t)
        dev = self.device
        d_action = th.empty((self.num_agent, 4), dtype=th.float32, device=dev)
        a = _lib.EnvBwdArgs()
        ref = self._tape_action_ref[t]
        a.tape_slab, a.action = self._tape[t].data_ptr(), (self._tape_actions[t] if ref is None else ref).data_ptr()
        keep = []
        for name, x, shape in (("d_obs", d_obs, (self.num_agent, 13)), ("d_reward", d_reward, (self.num_agent,))):
            if x is not None:
                x = x.to(dev, dtype=th.float32).reshape(shape).contiguous()
                keep.append(x)
                setattr(a, name, x.data_ptr())
        a.done, a.adj_slab, a.d_action = self._tape_done[t].data_ptr(), self._adj.data_ptr(), d_action.data_ptr()
        with th.cuda.device(dev):
            _lib.check(_lib.lib().vf_env_step_bwd(self._h, C.byref(a), self._stream()))
        return d_action

    def time_steps(self, action, iters=100, auto_reset=True):
        """mean device microseconds per fused env-step launch (HIP events on the current stream)"""
        N, dev = self.num_agent, self.device
        a = th.as_tensor(action, dtype=th.float32).to(dev).reshape(N, 4).contiguous()
        state = th.empty((N, 13), dtype=th.float32, device=dev)
        reward = th.empty(N, dtype=th.float32, device=dev)
        done = th.empty(N, dtype=th.bool, device=dev)
        o = self._out(state, reward, done)
        us = C.c_float(0)
        with th.cuda.device(dev):
            _lib.check(_lib.lib().vf_env_time_steps(self._h, _lib.ptr(a), C.byref(o), 1 if auto_reset else 0, int(iters),
                                                    self._stream(), C.byref(us)))
        self._qcache = None
        return float(us.value)

    # ------------------------------------------------------------------ reference surface
    def get_observation(self, indices=None, predicted_obs=None):
        return self._observations

    def get_full_observation(self, indice=None, predicted_obs=None):
        return self._observations

    def get_success(self):
        return self.success

    def get_failure(self):
        return th.zeros(self.num_agent, dtype=th.bool, device=self.device)

    def get_reward(self, predicted_obs=None):
        return self._reward

    def detach(self):
        """droneGymEnv.py:286-300: cut the autograd graph at the current state"""
        self.envs.detach()
        self.clear_tape()
        self._observations = TensorDict({k: (v.detach() if isinstance(v, th.Tensor) else v)
                                         for k, v in self._observations.items()})
        self._reward = self._reward.detach()

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            for ro in getattr(self, "_rollouts", {}).values():
                for g in ro["graphs"].values():
                    _lib.lib().vf_env_graph_destroy(g)
            self._rollouts = {}
            _lib.lib().vf_env_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_requires_grad(self, requires_grad: bool, horizon: int = 64):
        """droneGymEnv.py:628-633.  True: record the per-step checkpoints the adjoint kernel needs (up to
        `horizon` steps between two detach() calls) and return graph-attached obs / reward from step()."""
        self.requires_grad = bool(requires_grad)
        if requires_grad:
            if self._tape is None or self._tape.shape[0] < horizon:
                self.enable_tape(horizon)
            self._record_all = False  # autograd mode: only graph-attached steps are recorded
            self._token = th.zeros(1, device=self.device, requires_grad=True)
        else:
            self._tape = None

    def to(self, device):
        if th.device(device).type != "cuda":
            raise VisflyError("visfly_amd envs cannot move off the GPU")

    def env_is_wrapped(self):
        return False

    def get_attr(self, attr_name, indices=None):
        if indices is None:
            return getattr(self, attr_name)

    def __len__(self):
        return self.num_envs

    def __repr__(self):
        return (f"{self.__class__.__name__}(NumAgentPerScene={self.num_agent_per_scene}, NumScene={self.num_scene}, "
                f"tensorOut={self.tensor_output}, spawn={self.spawn_mode}, device={self.device})")

    # properties (droneGymEnv.py:477-571)
    reward = property(lambda s: s._reward)
    done = property(lambda s: s._done)
    info = property(lambda s: s._info)
    state = property(lambda s: s.envs.dynamics.state)
    position = property(lambda s: s.envs.dynamics.position)
    orientation = property(lambda s: s.envs.dynamics.orientation)
    velocity = property(lambda s: s.envs.dynamics.velocity)
    angular_velocity = property(lambda s: s.envs.dynamics.angular_velocity)
    direction = property(lambda s: s.envs.dynamics.direction)
    t = property(lambda s: s.envs.dynamics.t)
    full_state = property(lambda s: s.envs.dynamics.full_state)
    extend_state = property(lambda s: s.envs.dynamics.extend_state)
    visual = property(lambda s: False)
    sensor_obs = property(lambda s: {"IMU": s._imu_obs()})
    is_collision = property(lambda s: s.envs.is_collision)
    is_out_bounds = property(lambda s: s.envs.is_out_bounds)
    collision_point = property(lambda s: s.envs.collision_point)
    collision_vector = property(lambda s: s.envs.collision_vector)
    collision_dis = property(lambda s: s.envs.collision_dis)
    episode_done = property(lambda s: (s._query()["flags"] & F_EPISODE_DONE) != 0)
    success = property(lambda s: (s._query()["flags"] & F_SUCCESS) != 0)
    failure = property(lambda s: (s._query()["flags"] & F_FAILURE) != 0)
    _step_count = property(lambda s: s._query()["step_count"])
    _rewards = property(lambda s: s._query()["rewards"])
    _success = property(lambda s: s.success)
    _episode_done = property(lambda s: s.episode_done)
