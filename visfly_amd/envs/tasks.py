"""Task envs named in BASELINE.json: HoverEnv, NavigationEnv, RacingEnv (visual=False).

Constructor kwargs follow the reference (envs/HoverEnv.py:15-30, envs/NavigationEnv.py:27-41,
envs/RacingEnv.py:17-31); observation / reward / success definitions live in the fused kernel
(visfly_amd/csrc/vf_env_device.hpp) and are listed next to each class.
"""
import ctypes as C
from typing import Optional

import numpy as np
import torch as th

from . import spaces
from .. import _lib
from ..type import TensorDict
from .base import HOVER, NAV, RACING, DroneGymEnvsBase

_HOVER_SPAWN = {"state_generator": {"class": "Uniform", "kwargs": [
    {"position": {"mean": [1., 0., 1.5], "half": [1.0, 1.0, 0.5]}}]}}

_RACING_SPAWN = {"state_generator": {"class": "Union", "kwargs": [{"randomizers_kwargs": [
    {"class": "Uniform", "kwargs": {"position": {"mean": [2., 2., 1], "half": [.2, .2, 0.2]}}},
    {"class": "Uniform", "kwargs": {"position": {"mean": [6., 2., 1.5], "half": [.2, .2, 0.2]}}},
    {"class": "Uniform", "kwargs": {"position": {"mean": [6., -2., 1.5], "half": [.2, .2, 0.2]}}},
    {"class": "Uniform", "kwargs": {"position": {"mean": [2., 0., 1], "half": [.2, .2, 0.2]}}},
]}]}}

_RACING_GATES = [[4, 4, 1.], [8, 0, 2.], [5, -4, 1.], [1, -1, 1.]]


class HoverEnv(DroneGymEnvsBase):
    """obs {"state": (N,13)}; success == False; reward = 0.1 - |p-target|/90 - 1e-5|q-[1,0,0,0]|
    - 0.002|v| - 0.002|w| (envs/HoverEnv.py:62-94); default target [1,0,1.5], default spawn box
    mean [1,0,1.5] half [1,1,.5] (:32-41,:59)."""
    KIND = HOVER

    def __init__(self, num_agent_per_scene: int = 1, num_scene: int = 1, seed: int = 42, visual: bool = False,
                 requires_grad: bool = False, random_kwargs: Optional[dict] = None, dynamics_kwargs: Optional[dict] = None,
                 scene_kwargs: Optional[dict] = None, sensor_kwargs: Optional[list] = None, device="cuda",
                 target=None, max_episode_steps: int = 256, tensor_output: bool = False, **kw):
        super().__init__(num_agent_per_scene=num_agent_per_scene, num_scene=num_scene, seed=seed, visual=visual,
                         requires_grad=requires_grad, random_kwargs=_HOVER_SPAWN if random_kwargs is None else random_kwargs,
                         dynamics_kwargs=dynamics_kwargs, scene_kwargs=scene_kwargs, sensor_kwargs=sensor_kwargs,
                         device=device, max_episode_steps=max_episode_steps, tensor_output=tensor_output,
                         target=[1., 0., 1.5] if target is None else target, success_radius=0.5, **kw)
        self.target = th.ones((self.num_envs, 1), device=self.device) @ self.target.reshape(1, -1).to(self.device)


class NavigationEnv(DroneGymEnvsBase):
    """obs {"state": (N,13), "target": (N,3)}; success |p-target| <= 0.5; reward = progress toward the
    target, view-angle / attitude / speed penalties, obstacle terms and the success bonus
    (envs/NavigationEnv.py:63-99); default target [9,0,1] (:58)."""
    KIND = NAV

    def __init__(self, num_agent_per_scene: int = 1, num_scene: int = 1, seed: int = 42, visual: bool = False,
                 requires_grad: bool = False, random_kwargs: Optional[dict] = None, dynamics_kwargs: Optional[dict] = None,
                 scene_kwargs: Optional[dict] = None, sensor_kwargs: Optional[list] = None, device="cuda",
                 target=None, max_episode_steps: int = 256, tensor_output: bool = True, **kw):
        super().__init__(num_agent_per_scene=num_agent_per_scene, num_scene=num_scene, seed=seed, visual=visual,
                         requires_grad=requires_grad, random_kwargs=random_kwargs or {}, dynamics_kwargs=dynamics_kwargs,
                         scene_kwargs=scene_kwargs, sensor_kwargs=sensor_kwargs, device=device,
                         max_episode_steps=max_episode_steps, tensor_output=tensor_output,
                         target=[9., 0., 1.] if target is None else target, success_radius=0.5, **kw)
        self.target = th.ones((self.num_envs, 1), device=self.device) @ self.target.reshape(1, -1).to(self.device)
        self.observation_space["target"] = spaces.Box(low=-np.inf, high=np.inf, shape=(3,), dtype=np.float32)

    def _static_obs(self, i=None):
        return {"target": self.target if i is None else self.target[i]}


class HoverEnv2(HoverEnv):
    """HoverEnv with the scaled relative observation [(target - p)/10, q, v/10, w/10]
    (envs/HoverEnv.py:97-152); reward and success as HoverEnv.  The reference forces a depth sensor into
    sensor_kwargs (:115-119), which only matters with visual=True."""
    OBS_MODE = 1        # VF_OBS_HOVER2

    def _state_obs(self, raw):
        # tensor divisor: torch turns `x / python_scalar` on the GPU into x * (1/10), which is not the reference's
        # (CPU, true division) rounding; tensor / tensor is an IEEE division like the kernel's
        ten = th.full((1,), 10.0, device=raw.device)
        return th.hstack([(self.target - raw[:, 0:3]) / ten, raw[:, 3:7], raw[:, 7:10] / ten, raw[:, 10:13] / ten])


class NavigationEnv2(NavigationEnv):
    """obs {"state": [target - p, q, v, w], "collision_vector": (N,3)}; success |p-target| <= 0.5; failure =
    is_collision; reward = 0.02 (v_along - v_across toward the target) - 0.001 |w| + success
    (envs/NavigationEnv.py:102-224); default target [14,0,1], default spawn U(mean [9,0,1.5], half [8,6,1])."""
    OBS_MODE = 2        # VF_OBS_NAV2
    REWARD_MODE = 1     # VF_REWARD_NAV2
    _STATIC_OBS_CONST = False   # collision_vector is queried per step

    def __init__(self, *a, random_kwargs=None, target=None, **kw):
        spawn = {"state_generator": {"class": "Uniform", "kwargs": [
            {"position": {"mean": [9., 0., 1.5], "half": [8.0, 6., 1.]}}]}} if random_kwargs is None else random_kwargs
        super().__init__(*a, random_kwargs=spawn, target=[14., 0., 1.] if target is None else target, **kw)
        self.max_sense_radius = 10
        del self.observation_space.spaces["target"]
        self.observation_space["collision_vector"] = spaces.Box(low=-np.inf, high=np.inf, shape=(3,), dtype=np.float32)

    def _state_obs(self, raw):
        return th.hstack([self.target - raw[:, 0:3], raw[:, 3:13]])

    def _static_obs(self, i=None):
        cv = self.collision_vector
        return {"collision_vector": cv if i is None else cv[i]}


class RacingEnv(DroneGymEnvsBase):
    """obs {"state": (N,13), "gate": (N,) next-gate index}; 4 gates, pass radius 0.3; passing advances
    the gate (mod 4) and pays +20 on top of the hover-style reward toward the next gate; gate chosen at
    reset from the spawn position; Union-of-4-Uniform spawn (envs/RacingEnv.py:33-70,87-98,142-215)."""
    KIND = RACING

    def __init__(self, num_agent_per_scene: int = 1, num_scene: int = 1, seed: int = 42, visual: bool = False,
                 requires_grad: bool = False, random_kwargs: Optional[dict] = None, dynamics_kwargs: Optional[dict] = None,
                 scene_kwargs: Optional[dict] = None, sensor_kwargs: Optional[list] = None, device="cuda",
                 target=None, max_episode_steps: int = 256, tensor_output: bool = True, latent_dim=None, gates=None, **kw):
        # the reference ignores a caller's random_kwargs and always uses the 4-box union (RacingEnv.py:32-70)
        super().__init__(num_agent_per_scene=num_agent_per_scene, num_scene=num_scene, seed=seed, visual=visual,
                         requires_grad=requires_grad, random_kwargs=_RACING_SPAWN, dynamics_kwargs=dynamics_kwargs,
                         scene_kwargs=scene_kwargs, sensor_kwargs=sensor_kwargs, device=device,
                         max_episode_steps=max_episode_steps, tensor_output=tensor_output,
                         success_radius=0.3, gates=_RACING_GATES if gates is None else gates, **kw)
        self._next_target_num = 2
        self.success_r = 20
        self.observation_space["gate"] = spaces.Box(low=0, high=len(_RACING_GATES), shape=(1,), dtype=np.int32)
        self.observation_space["state"] = spaces.Box(
            low=-np.inf, high=np.inf,
            shape=(3 * (self._next_target_num - 1) + self.observation_space["state"].shape[0],), dtype=np.float32)

    # The gate index inside the observation step() / reset() RETURN is not always the env's current one (all of it pinned by the
    # repaired-oracle fixtures env_racing / env_racing2):
    #  * DroneGymEnvsBase.step refreshes the observation BEFORE get_success() advances the gate of an agent that just passed one
    #    (droneGymEnv.py:161-166): the returned entry is the gate the agent was flying to at the START of the step;
    #  * unless some agent ended its episode in this step: examine() -> reset_agent_by_id -> get_full_observation(indices)
    #    ignores its indices and rebuilds EVERY agent's observation (:347,460-469), now with the advanced / newly chosen gates;
    #  * RacingEnv.reset builds its observation inside super().reset(), before it re-chooses the gates (RacingEnv.py:165-170):
    #    the returned entry is the previous episode's (zeros on the first reset).
    # `obs_gate_exact = False` (set by trainers whose policy does not read "gate") skips the two small launches this costs per
    # step and returns the current index.
    obs_gate_exact = True
    _g_obs = None
    _STATIC_OBS_CONST = False

    def _static_obs(self, i=None):
        g = self._gate if self._g_obs is None else self._g_obs
        return {"gate": g if i is None else g[i]}

    def _full_obs(self, state, raw=False):
        self._last_raw = state
        return super()._full_obs(state, raw)

    def _step_no_grad(self, _action, is_test=False, **kw):
        self._g_obs = None
        if not self.obs_gate_exact:
            return super()._step_no_grad(_action, is_test=is_test, **kw)
        prev = self._gate.clone()
        _, reward, done, info = super()._step_no_grad(_action, is_test=is_test, **kw)
        self._g_obs = prev if is_test else th.where(self._done.any(), self._gate, prev)
        self._observations = obs = self._full_obs(self._last_raw)
        return (obs if self.tensor_output else self._format_obs(obs)), reward, done, info

    def step_n(self, *a, **kw):
        self._g_obs = None          # step_n returns raw rows only; get_observation() afterwards reports the current gates
        return super().step_n(*a, **kw)

    def get_observation(self, indices=None, predicted_obs=None):
        # the reference recomputes the observation from its attributes on every call (RacingEnv.py:118-134): current gates
        self._g_obs = None
        self._observations = self._full_obs(self._last_raw)
        return self._observations

    def _reset_kernel(self, idx, fs):
        super()._reset_kernel(idx, fs)
        self._gate.copy_(self._query()["gate"])     # in place: the step kernel holds this buffer's address
        self._g_obs = None

    def _terminal_static_obs(self, i):
        return {"gate": self._terminal_gate[i]}

    def _extra_info(self):
        # gates passed in the episode that just ended, written by the step kernel where done (RacingEnv.py:113-116)
        return {"past_gate": self._ep_past_gates}

    def reset(self, state=None, obs=None, **kw):
        stale = self._gate.clone()
        out = super().reset(state)
        if not self.obs_gate_exact:
            return out
        self._g_obs = stale
        self._observations = o = self._full_obs(self.envs.dynamics.state, raw=True)
        return self._format_obs(o)

    _next_target_i = property(lambda s: s._query()["gate"])
    _past_targets_num = property(lambda s: s._query()["past_gates"])


class RacingEnv2(RacingEnv):
    """RacingEnv with the gate-relative observation (envs/RacingEnv.py:218-267): "state" = [(next `_next_target_num` gates -
    p) / max_sense_radius (6), q (4), v / 10 (3), w / 10 (3)] = 16 columns, "gate" = next-gate index as (N,1).  Dynamics, reward,
    gate bookkeeping and re-spawn are RacingEnv's (the step kernel); the observation rows are formed from the kernel's raw state rows
    by ONE launch behind it (vf_race_obs, csrc/vf_obs.hip; the reference's torch expressions remain as the host path for single
    terminal rows) (NEXT tier; step_n() is refused).
    requires_grad=True (r05): the observation is linear in the raw state row for a given gate index (which carries no gradient), so
    its adjoint is four column operations in front of the step kernel's adjoint (``backward_step``); pinned by the gradient fixture
    ``bptt_racing2_thrust`` (the reference's autograd over its own RacingEnv2.get_observation).
    r06: the trainers' persistent launches (BPTT / SHAC: vf_bptt_rollout / vf_bptt_reverse) form and differentiate the 16 columns
    themselves (vf_env_cfg.obs_mode = VF_OBS_RACE2: the row of the agent's CURRENT gate, which is what a trainer's policy reads --
    ``obs_gate_exact`` False); step() keeps the launch behind the step kernel, whose gate index is the batch-wide rule."""
    _HOST_OBS = True      # step() / step_n(): the rows are formed behind the step kernel
    OBS_MODE = 3          # VF_OBS_RACE2: ... and inside the persistent launches
    _OBS_W = 16

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.max_sense_radius = 10                                                     # droneGymEnv.py:69
        g = kw.get("gates")
        self.targets = th.as_tensor(_RACING_GATES if g is None else g, dtype=th.float32, device=self.device)
        self._gate_prev = th.zeros_like(self._gate)
        self.enable_done_list()        # vf_race_obs reads the step's done count (which gate index the returned rows use)

    # `targets` (the gates the observation is relative to) is what vf_race_obs reads through a host copy: assigning it refreshes the
    # copy, so that the one-launch rows, the torch expressions (single / terminal rows) and the caller agree.  (The step
    # kernel's own gates -- reward, passing, re-spawn -- are the constructor's `gates`, as in RacingEnv.)
    @property
    def targets(self):
        return self._targets

    @targets.setter
    def targets(self, v):
        self._targets = v
        if v is not None:
            self._targets = th.as_tensor(v, dtype=th.float32).to(getattr(self, "device", "cpu"))
            self._gates_host = (C.c_float * (3 * len(self._targets)))(*[float(x) for x in self._targets.cpu().reshape(-1).tolist()])

    def _after_persistent_launch(self):
        self._last_raw = None          # get_observation() re-reads the raw rows from the slab
        self._g_obs = None

    def get_observation(self, indices=None, predicted_obs=None):
        if self._last_raw is None:
            self._last_raw = self.envs.dynamics.state.detach()
        return super().get_observation(indices, predicted_obs)

    def _race_rows(self, raw, gate, gate_prev=None, mode=0, gate_out=None):
        """vf_race_obs: the observation rows in one launch -> (N, 16) tensor (mode: include/visfly_amd.h)"""
        N = raw.shape[0]
        out = th.empty((N, 3 * self._next_target_num + 10), dtype=th.float32, device=raw.device)
        dl = getattr(self, "_done_list", None)
        with th.cuda.device(raw.device):
            _lib.check(_lib.lib().vf_race_obs(raw.data_ptr(), gate.data_ptr(), None if gate_prev is None else gate_prev.data_ptr(),
                                              None if dl is None else dl[1].data_ptr(), mode, self._gates_host, len(self.targets),
                                              self._next_target_num, float(self.max_sense_radius), out.data_ptr(),
                                              None if gate_out is None else gate_out.data_ptr(), N, self._stream()))
        return out

    def _step_no_grad(self, _action, is_test=False, **kw):
        """RacingEnv's step with the returned rows formed by ONE launch behind the step kernel (vf_race_obs picks the gate index by
        RacingEnv's rule from the step's done count) instead of a clone, an any(), a where() and the dozen launches of _race_state"""
        if not self.obs_gate_exact or not self.tensor_output or (getattr(self, "_done_list", None) is None and not is_test):
            return super()._step_no_grad(_action, is_test=is_test, **kw)      # (enable_done_list(False): the torch path needs no done count)
        self._g_obs = None
        self._gate_prev.copy_(self._gate)
        self._defer_rows = True                    # the base step's own _full_obs call only records the raw rows
        try:
            _, reward, done, info = super(RacingEnv, self)._step_no_grad(_action, is_test=is_test, **kw)
        finally:
            self._defer_rows = False
        g = th.empty_like(self._gate)
        state = self._race_rows(self._last_raw, self._gate, self._gate_prev, 1 if is_test else 2, g)
        self._g_obs = g
        self._observations = obs = TensorDict({"state": state, "gate": g.unsqueeze(1)})
        return obs, reward, done, info

    def _race_state(self, raw, gate):
        if (raw.is_cuda and raw.dtype == th.float32 and raw.dim() == 2 and raw.shape[1] == 13 and raw.is_contiguous() and raw.shape[0] > 1
                and gate.dtype == th.int32 and gate.is_contiguous() and gate.shape[0] == raw.shape[0]):
            return self._race_rows(raw, gate)
        idx = th.stack([gate + i for i in range(self._next_target_num)]).T % len(self.targets)    # RacingEnv.py:254
        rel = (self.targets[idx.long()] - raw[:, 0:3].unsqueeze(1)).reshape(raw.shape[0], -1)       # :255-256
        # divisors as device tensors: torch's GPU kernels turn `x / python_scalar` into x * (1 / scalar), one ulp off the CPU
        # reference's IEEE division
        ten = th.full((1,), 10.0, device=raw.device)
        return th.hstack([rel / th.full((1,), float(self.max_sense_radius), device=raw.device), raw[:, 3:7],
                          raw[:, 7:10] / ten, raw[:, 10:13] / ten])                                    # :257-262

    def backward_step(self, t: int, d_obs=None, d_reward=None):
        """d_obs (N, 16): gradient w.r.t. the observation rows step t returned -> the raw state row's (N, 13), then RacingEnv's adjoint.
        obs = [(g0 - p) / R, (g1 - p) / R, q, v / 10, w / 10] (RacingEnv.py:254-262): dp = -(d[0:3] + d[3:6]) / R, dq = d[6:10],
        dv = d[10:13] / 10, dw = d[13:16] / 10"""
        if d_obs is not None:
            n3 = 3 * self._next_target_num            # columns of the gate-relative block (6 for the reference's two gates)
            d = d_obs.to(self.device, dtype=th.float32).reshape(self.num_agent, n3 + 10)
            # divisors as device tensors: IEEE divisions like the persistent launch's race2_obs_bwd (a python scalar becomes x * (1 / s))
            R, ten = th.full((1,), float(self.max_sense_radius), device=self.device), th.full((1,), 10.0, device=self.device)
            dp = d[:, 0:3]
            for j in range(1, self._next_target_num):
                dp = dp + d[:, 3 * j:3 * j + 3]
            d_obs = th.cat([-dp / R, d[:, n3:n3 + 4], d[:, n3 + 4:n3 + 7] / ten, d[:, n3 + 7:n3 + 10] / ten], dim=1)
        return super().backward_step(t, d_obs, d_reward)

    def _full_obs(self, state, raw=False):          # RacingEnv's rules for WHICH gate index the returned rows use apply unchanged
        if getattr(self, "_defer_rows", False):        # inside _step_no_grad: the rows are formed after the step kernel, from these
            self._last_raw = state
            return TensorDict({"state": state, "gate": self._gate.unsqueeze(1)})
        g = self._gate if self._g_obs is None else self._g_obs
        if state.shape[-1] == 16:      # already the observation rows (step() with requires_grad: the graph-attached copy of what
            return TensorDict({"state": state, "gate": g.unsqueeze(1).clone()})      # _step_no_grad built from the raw rows)
        self._last_raw = state
        return TensorDict({"state": self._race_state(state, g), "gate": g.unsqueeze(1).clone()})      # :264-267

    def _terminal_state_obs(self, tobs, i):
        raw = tobs[i].reshape(1, 13).to(self.device)
        return self._race_state(raw, self._terminal_gate[i].reshape(1))[0]

    def _terminal_state_rows(self):
        return self._race_state(self._terminal_obs, self._terminal_gate)

    def _terminal_static_obs(self, i):
        return {"gate": self._terminal_gate[i].reshape(1)}
