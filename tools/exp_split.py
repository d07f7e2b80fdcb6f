"""A/B the one-thread-per-agent kernel against the two-wave split (VISFLY_AMD_SPLIT=0/1), interleaved rounds"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys
sys.path.insert(0, %r)
import torch
from visfly_amd.envs import HoverEnv
from visfly_amd import Dynamics
kw = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
out = {}
for N in (64, 16384, 32768, 65536, 131072, 262144, 524288, 1048576):
    env = HoverEnv(num_agent_per_scene=N, dynamics_kwargs=kw, device="cuda:0", tensor_output=True, max_episode_steps=256)
    env.reset()
    a = (torch.rand((N, 4), device="cuda") * 2 - 1) * 0.02 + torch.tensor([-1 / 3, 0, 0, 0], device="cuda")
    env.time_steps(a, 50)
    e = min(env.time_steps(a, 200) for _ in range(5))
    d = env.envs.dynamics
    dd = min(d.time_steps(a, 200) for _ in range(3))
    out[N] = (round(e, 2), round(dd, 2))
print(out)
''' % root
for rnd in range(2):
    for split in ("0", "1"):
        env = dict(os.environ, VISFLY_AMD_SPLIT=split)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        print("split=" + split, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:])
