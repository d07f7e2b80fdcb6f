"""k_mlp_forward_chain (32 rows per wave) vs k_mlp_forward_chain16 (16 rows per wave) in isolation: the BPTT actor (Hover / Racing
state 13 -> [128, 64] -> pi [64, 64] -> 4, action head fused) at M rows, HIP events over 200 launches.
    VISFLY_AMD_MLP_CHAIN16=0|1 python tools/exp_chain16.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from visfly_amd.ppo import MlpPolicy
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = th.device("cuda", 0)
pol = MlpPolicy({"state": 13}, {"state": [128, 64]}, [64, 64], [64, 64], dev, seed=1)
obs = {"state": th.randn((M, 13), device=dev)}
eps = th.randn((M, 4), device=dev)
act = th.empty((M, 4), device=dev)
for save in (True,):
    assert pol.forward_act(obs, eps, act, slot=0)
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(3):
        e0.record()
        for _ in range(200):
            pol.forward_act(obs, eps, act, slot=0)
        e1.record(); th.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 200)
    print(f"M={M} chain16={os.environ.get('VISFLY_AMD_MLP_CHAIN16', 'auto')}: forward_act {best:.2f} us per launch (incl. host pacing)")
