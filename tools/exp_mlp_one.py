"""a few fused forward / backward launches of the Nav actor-critic at M rows, for rocprofv3 --pmc passes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd.ppo import MlpPolicy
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
DEV = "cuda:0"
pol = MlpPolicy({"state": 13, "target": 3}, {"state": [128, 64], "target": [128, 64]}, [64, 64], [64, 64], DEV, seed=9)
pol.lazy_pack = True
obs = {"state": torch.randn((M, 13), device=DEV), "target": torch.randn((M, 3), device=DEV)}
dm, dv = torch.randn((M, 4), device=DEV), torch.randn(M, device=DEV)
for _ in range(n):
    pol.forward(obs)
    pol.backward(dm, dv, None)
torch.cuda.synchronize()
