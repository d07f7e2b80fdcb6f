"""Task envs named in BASELINE.json: HoverEnv, NavigationEnv, RacingEnv (visual=False).

Constructor kwargs follow the reference (envs/HoverEnv.py:15-30, envs/NavigationEnv.py:27-41,
envs/RacingEnv.py:17-31); observation / reward / success definitions live in the fused kernel
(visfly_amd/csrc/vf_env_device.hpp) and are listed next to each class.
"""
from typing import Optional

import numpy as np
import torch as th

from . import spaces
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

    def _static_obs(self, i=None):
        return {"gate": self._gate if i is None else self._gate[i]}

    def _reset_kernel(self, idx, fs):
        super()._reset_kernel(idx, fs)
        self._gate.copy_(self._query()["gate"])     # in place: the step kernel holds this buffer's address

    def _terminal_static_obs(self, i):
        return {"gate": self._terminal_gate[i]}

    def _extra_info(self):
        # gates passed in the episode that just ended, written by the step kernel where done (RacingEnv.py:113-116)
        return {"past_gate": self._ep_past_gates}

    def reset(self, state=None, obs=None, **kw):
        return super().reset(state)

    _next_target_i = property(lambda s: s._query()["gate"])
    _past_targets_num = property(lambda s: s._query()["past_gates"])
