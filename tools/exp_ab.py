"""A/B two builds of libvisfly_amd.so on the env-step kernel (interleaved rounds, one process per variant)"""
import sys, os, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import torch
from visfly_amd import _build
_build.LIB = sys.argv[1]
from visfly_amd import _lib
_lib.LIB = sys.argv[1]
from visfly_amd.envs import HoverEnv
kw = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
out = {}
for N in (64, 65536, 1048576):
    env = HoverEnv(num_agent_per_scene=N, dynamics_kwargs=kw, device="cuda:0", tensor_output=True, max_episode_steps=256)
    env.reset()
    a = (torch.rand((N, 4), device="cuda") * 2 - 1) * 0.02 + torch.tensor([-1 / 3, 0, 0, 0], device="cuda")
    env.time_steps(a, 50)
    out[N] = min(env.time_steps(a, 200) for _ in range(5))
print(out)
''' % root
libs = sys.argv[1:]
for rnd in range(2):
    for lib in libs:
        r = subprocess.run([sys.executable, "-c", code, lib], capture_output=True, text=True)
        print(os.path.basename(lib), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
