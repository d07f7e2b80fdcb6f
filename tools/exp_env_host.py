"""is the bench loop host- or GPU-bound?  host time to ENQUEUE env.step vs time until the GPU is done"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd.envs import HoverEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
kw = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
env = HoverEnv(num_agent_per_scene=N, dynamics_kwargs=kw, device="cuda:0", tensor_output=True, max_episode_steps=256)
env.reset()
a = ((torch.rand((N, 4), device="cuda") * 2 - 1) * 0.02 + torch.tensor([-1 / 3, 0, 0, 0], device="cuda")).contiguous()
for _ in range(200): env.step(a)
torch.cuda.synchronize()
n = 3000
t0 = time.perf_counter()
for _ in range(n): env.step(a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"N={N}: host enqueue {1e6 * (t1 - t0) / n:.2f} us/step   until GPU done {1e6 * (t2 - t0) / n:.2f} us/step   kernel {env.time_steps(a, 300):.2f} us")
