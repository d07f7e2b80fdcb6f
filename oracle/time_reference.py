"""Times the ACTUAL reference (PyTorch CPU, imported from /root/reference through the stubs of gen_golden.py) in the build
container: Dynamics.step and HoverEnv.step at the bench shape.  Test infrastructure, runs only where /root/reference exists; the
output is committed as profiles/r02_reference_cpu_here.txt (the reference cannot travel to the GPU box).
    python oracle/time_reference.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402
import torch as th  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
KW = dict(action_type="bodyrate", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True, integrator="euler")
print(f"torch {th.__version__}, {th.get_num_threads()} threads, {os.cpu_count()} cpus; N = {N}")
D = G.import_dynamics()
d = D(num=N, **KW)
d.reset()
a = (th.rand(N, 4) * 2 - 1) * 0.02 + th.tensor([-1 / 3, 0, 0, 0])
for _ in range(2):
    d.step(a)
t0 = time.perf_counter()
K = 6
for _ in range(K):
    d.step(a)
el = time.perf_counter() - t0
print(f"reference Dynamics.step : {el / K * 1e3:8.1f} ms per step = {N * K / el:.3e} agent-steps/s")
H, _, _ = G.import_envs()
env = H(num_agent_per_scene=min(N, 4096), num_scene=1, seed=42, visual=False, dynamics_kwargs=dict(KW), device="cpu", max_episode_steps=256)
env.tensor_output = True
n = env.num_agent
env.reset()
a = (th.rand(n, 4) * 2 - 1) * 0.02 + th.tensor([-1 / 3, 0, 0, 0])
for _ in range(2):
    env.step(a)
t0 = time.perf_counter()
for _ in range(K):
    env.step(a)
el = time.perf_counter() - t0
print(f"reference HoverEnv.step : {el / K * 1e3:8.1f} ms per step = {n * K / el:.3e} agent-steps/s  (N = {n}: reset() of 65 536 agents alone takes ~11 s in the reference)")
