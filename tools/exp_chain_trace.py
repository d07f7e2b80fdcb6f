import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from visfly_amd.ppo import MlpPolicy
from visfly_amd import _lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = th.device("cuda", 0)
pol = MlpPolicy({"state": 13}, {"state": [128, 64]}, [64, 64], [64, 64], dev, seed=1)
obs = {"state": th.randn((M, 13), device=dev)}
eps = th.randn((M, 4), device=dev); act = th.empty((M, 4), device=dev)
for _ in range(20): pol.forward_act(obs, eps, act, slot=0)
th.cuda.synchronize()
L = C.CDLL(_lib.lib()._name)
out = (C.c_longlong * 64)()
L.vf_debug_chain_trace(out)
for w in range(2):
    t = [out[w * 32 + k] for k in range(32)]
    base = t[0]
    print("wave", "first" if w == 0 else "last", [(k, t[k] - base) for k in range(32) if t[k]])
