"""Evaluation rollout of a trained policy: the numeric part of the reference's ``TestBase.test``
(utils/evaluate.py:57-129) -- reset, deterministic actions until every agent has finished one episode, per-step
records, mean episode return / length.  Figures and videos (``draw`` / ``play`` / ``save_video``, matplotlib / cv2)
are outside the hot path and not reproduced.
"""
from typing import Dict, List

import torch as th


class TestBase:
    """same constructor idea as utils/evaluate.py:24-55: a model (anything with ``predict(obs, deterministic=True)``:
    visfly_amd.ppo.PPO / BPTT / SHAC) and the env to run it on"""

    def __init__(self, model=None, env=None, name=None):
        self.model, self.env, self.name = model, (env if env is not None else model.env), name
        self.obs_all: List[Dict[str, th.Tensor]] = []
        self.state_all, self.reward_all, self.action_all, self.t, self.collision_all = [], [], [], [], []
        self.eq_r: List[float] = []
        self.eq_l: List[float] = []

    def test(self, max_steps: int = 100000):
        """-> (mean episode return, mean episode length) over the FIRST episode of every agent (:119-129)"""
        env, model = self.env, self.model
        tensor_output, env.tensor_output = env.tensor_output, True
        obs = env.reset()
        self.obs_all.append(obs)
        self.state_all.append(env.extend_state.clone())
        self.t.append(env.t.clone())
        N = env.num_envs
        open_mask = th.ones(N, dtype=th.bool, device=env.device)
        ret = th.zeros(N, device=env.device)
        length = th.zeros(N, device=env.device)
        for _ in range(max_steps):
            action = model.predict(obs, deterministic=True)
            action = action[0] if isinstance(action, tuple) else action
            obs, reward, done, _info = env.step(action, is_test=True)
            self.collision_all.append({"col_dis": env.collision_dis, "is_col": env.is_collision, "col_pt": env.collision_point})
            self.reward_all.append(reward)
            self.action_all.append(action)
            self.state_all.append(env.extend_state.clone())
            self.obs_all.append(obs)
            self.t.append(env.t.clone())
            # info[i]["episode"] of the reference = return / length of the episode that just ended: kept on the device here
            fin = done & open_mask
            ret = th.where(fin, env._ep_return, ret)
            length = th.where(fin, env._ep_length.to(length.dtype), length)
            open_mask &= ~done
            if not bool(open_mask.any()):
                break
        env.tensor_output = tensor_output
        closed = ~open_mask
        self.eq_r, self.eq_l = ret[closed].tolist(), length[closed].tolist()
        mean_r = float(ret[closed].mean()) if bool(closed.any()) else float("nan")
        mean_l = float(length[closed].mean()) if bool(closed.any()) else float("nan")
        return mean_r, mean_l
