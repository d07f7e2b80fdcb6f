"""experiment: fused-step kernel time vs number of agents (latency- or throughput-bound?)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd import Dynamics

kw = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
for N in [64, 4096, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 4194304]:
    d = Dynamics(num=N, device="cuda:0", **kw)
    a = (torch.rand((N, 4), device="cuda") * 2 - 1) * 0.02 + torch.tensor([-1 / 3, 0, 0, 0], device="cuda")
    d.time_steps(a, 50)
    us = min(d.time_steps(a, 200) for _ in range(3))
    print(f"N={N:8d}  {us:8.2f} us/launch  {N / us * 1e6:.3e} agent-steps/s  {244 * N / us / 1e3:8.1f} GB/s algorithmic")
