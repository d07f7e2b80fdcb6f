"""PPO.train() with the minibatches read through the permutation slice (vf_ppo_loss_cfg.row_index, ABI 9: no shuffled copy of the rollout
buffer per epoch) vs with the per-epoch shuffled copy (vf_gather_rows): bench.py's PPO workload (25 600 agents x 256 steps, batch 25 600,
5 epochs = 1280 optimiser steps per call), alternating, 3 x 6 calls each.  Result: profiles/r05_side_streams.txt section 4."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from visfly_amd.envs import NavigationEnv
from visfly_amd.ppo import PPO

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = "cuda:0"
dyn = dict(action_type="bodyrate", ori_output_type="quaternion", dt=0.0025, ctrl_dt=0.02, integrator="euler", drag_random=0.0)
env = NavigationEnv(num_agent_per_scene=25600, seed=1, device=dev, max_episode_steps=256, tensor_output=True, dynamics_kwargs=dict(dyn))
ppo = PPO(env, n_steps=256, batch_size=25600, n_epochs=5, learning_rate=1e-4, seed=0, policy_kwargs=dict(activation_fn="relu"))
ppo.collect_rollouts()
rows = []
for rep in range(3):
    for flag in (False, True):
        ppo.index_minibatches = flag
        ppo.train()
        th.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            ppo.train()
        th.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        rows.append((flag, dt * 1e3))
        print(f"rep {rep}  index_minibatches={flag!s:5}  train() {dt * 1e3:8.3f} ms", flush=True)
for flag in (False, True):
    v = sorted(r[1] for r in rows if r[0] == flag)
    print(f"index_minibatches={flag!s:5}  median {v[1]:8.3f} ms  min {v[0]:8.3f} ms   {v[1] * 1e3 / 1280:6.2f} us per optimiser step")
