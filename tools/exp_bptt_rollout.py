"""device time of the forward half of a BPTT horizon: persistent launch (vf_bptt_rollout) vs launch by launch, and of the whole update"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd.bptt import BPTT
from visfly_amd.envs import RacingEnv
N, H = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 64
dkw = dict(action_type="thrust", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
for fused in (True, False, True, False):
    env = RacingEnv(num_agent_per_scene=N, seed=42, dynamics_kwargs=dkw, device="cuda:0", max_episode_steps=256, requires_grad=True, tensor_output=True)
    algo = BPTT(env, horizon=H, gamma=0.99, learning_rate=1e-3, seed=0)
    algo.fused_rollout = fused
    for _ in range(3):
        algo._update()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    e[0].record()
    for _ in range(10):
        algo._update()
    e[1].record()
    torch.cuda.synchronize()
    ms = e[0].elapsed_time(e[1]) / 10
    print(f"N={N} H={H} fused_rollout={fused}: {ms:.3f} ms per update -> {N * H / ms * 1e3:.3e} env-steps/s")
    env.close()
