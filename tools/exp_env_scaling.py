"""fused HoverEnv.step launch time vs number of agents: what does a SIMD gain when 2 / 4 / 16 waves share it?  (profiles/r04_env_quad.txt:
the term of the four-lanes-per-agent budget that no instruction count gives -- how much of a lone wave's 44 % of parked time other
waves on the same SIMD fill)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd.envs import HoverEnv

kw = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
base = None
for N in (16384, 32768, 65536, 131072, 262144, 524288, 1048576):
    env = HoverEnv(num_agent_per_scene=N, dynamics_kwargs=kw, device="cuda:0", tensor_output=True, max_episode_steps=256)
    env.reset()
    a = (torch.rand((N, 4), device="cuda") * 2 - 1) * 0.02 + torch.tensor([-1 / 3, 0, 0, 0], device="cuda")
    env.time_steps(a, 60)
    us = min(env.time_steps(a, 200) for _ in range(5))
    if N == 65536:
        base = us
    print(f"N={N:8d}  waves/SIMD {N / 65536:5.2f}  {us:8.2f} us/launch  {us * 65536 / N:6.2f} us per 65 536 agents  {N / us * 1e6:.3e} agent-steps/s")
    env.close()
