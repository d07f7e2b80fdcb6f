"""k_dyn_step alone vs the fused k_env_step on the same slab (HIP events around 300 launches each), standalone Dynamics too"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd.envs import HoverEnv
from visfly_amd.dynamics import Dynamics
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
kw = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
env = HoverEnv(num_agent_per_scene=N, dynamics_kwargs=kw, device="cuda:0", tensor_output=True, max_episode_steps=256)
env.reset()
a = ((torch.rand((N, 4), device="cuda") * 2 - 1) * 0.02 + torch.tensor([-1 / 3, 0, 0, 0], device="cuda")).contiguous()
dyn = env.envs.dynamics
for rep in range(3):
    print("env", round(env.time_steps(a, iters=300), 3), "dyn(embedded)", round(dyn.time_steps(a, iters=300), 3))
d2 = Dynamics(num=N, device=torch.device("cuda:0"), **kw)
d2.reset()
for rep in range(3):
    print("dyn(standalone)", round(d2.time_steps(a, iters=300), 3))
print("finite", bool(torch.isfinite(d2.state).all()), bool(torch.isfinite(dyn.state).all()))
