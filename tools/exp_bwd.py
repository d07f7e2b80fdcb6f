"""fused whole-network backward (vf_mlp_backward) vs the layer-by-layer kernels, Nav actor-critic"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd.ppo import MlpPolicy
DEV = "cuda:0"
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
pol = MlpPolicy({"state": 13, "target": 3}, {"state": [128, 64], "target": [128, 64]}, [64, 64], [64, 64], DEV, seed=9)
for M in (4096, 25600, 32768, 65536):
    obs = {"state": torch.randn((M, 13), device=DEV), "target": torch.randn((M, 3), device=DEV)}
    dm, dv, dl = torch.randn((M, 4), device=DEV), torch.randn(M, device=DEV), torch.randn(4, device=DEV)
    pol.forward(obs)
    out = []
    for fused in (False, True):
        pol.fused_backward = fused
        out.append(timeit(lambda: pol.backward(dm, dv, dl)))
    fl = 2 * 2 * M * 43350 / 1e6   # dX + dW flops
    print(f"M={M:6d}: layerwise {out[0]:7.1f} us   fused {out[1]:7.1f} us ({fl / out[1] / 1e6 * 1e3:5.1f} TF/s)")
