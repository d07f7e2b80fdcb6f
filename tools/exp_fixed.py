"""how much of the kernel time is fixed (launch, load burst, dirty write-back) vs the sub-step compute:
time the dynamics kernel with 1, 2, 4, 8 sub-steps per control interval"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd import Dynamics
for N in (64, 65536):
    for sub in (1, 2, 4, 8, 16):
        d = Dynamics(num=N, device="cuda:0", action_type="bodyrate", dt=0.02 / sub, ctrl_dt=0.02)
        a = (torch.rand((N, 4), device="cuda") * 2 - 1) * 0.02 + torch.tensor([-1 / 3, 0, 0, 0], device="cuda")
        d.time_steps(a, 50)
        print(f"N={N:6d} sub-steps={sub:2d}: {min(d.time_steps(a, 200) for _ in range(4)):7.2f} us/launch")
