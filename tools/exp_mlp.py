"""fused whole-network forward vs layer-by-layer launches (torch.cuda events)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd.ppo import MlpPolicy
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
pol = MlpPolicy({"state": 13, "target": 3}, {"state": [128, 64], "target": [128, 64]}, [64, 64], [64, 64], "cuda:0")
for M in (4096, 25600, 32768, 65536):
    obs = {"state": torch.randn((M, 13), device="cuda"), "target": torch.randn((M, 3), device="cuda")}
    pol.fused = True
    a = timeit(lambda: pol.forward(obs, save_activations=False))
    b = timeit(lambda: pol.forward(obs, save_activations=True))
    pol.fused = False
    c = timeit(lambda: pol.forward(obs))
    fl = 86.7e3 * M
    print(f"M={M:6d}: fused(infer) {a:7.1f} us ({fl/a/1e6:5.1f} TF/s)  fused(save) {b:7.1f} us  layerwise {c:7.1f} us")
