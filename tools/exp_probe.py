"""phase breakdown (shader clocks of block 0) of the fused MLP forward / backward; builds a -DVF_PROBE library"""
import ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from visfly_amd import _build, _lib
so = os.path.join(root, "visfly_amd", "csrc", "libvf_probe.so")
if not os.path.exists(so) or "--rebuild" in sys.argv or os.path.getmtime(so) < os.path.getmtime(os.path.join(root, "visfly_amd", "csrc", "vf_ppo.hip")):
    _build.build(force=True, extra_flags=["-DVF_PROBE"], out=so)
_lib.LIB = so
from visfly_amd.ppo import MlpPolicy
lib = _lib.lib()
raw = C.CDLL(so)
raw.vf_probe_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int32]
def read(reset=1):
    buf = (C.c_ulonglong * 32)()
    assert raw.vf_probe_read(buf, reset) == 0
    return list(buf)
DEV = "cuda:0"
pol = MlpPolicy({"state": 13, "target": 3}, {"state": [128, 64], "target": [128, 64]}, [64, 64], [64, 64], DEV, seed=9)
FWD = {0: "tile-top barrier", 1: "obs -> LDS", 2: "layer-top barrier", 3: "zero + W loads/ds_write", 4: "barrier after W",
       5: "MFMA sweep", 6: "epilogue"}
BWD = {8: "item 0 staging", 9: "issue next item + B frag", 13: "bias + dW MFMA", 14: "dX MFMA + store", 10: "partials (layer end)",
       15: "barrier (buffer consumed)", 11: "park next item", 12: "barrier (buffer ready)"}
for M in (64, 25600):
    obs = {"state": torch.randn((M, 13), device=DEV), "target": torch.randn((M, 3), device=DEV)}
    dm, dv, dl = torch.randn((M, 4), device=DEV), torch.randn(M, device=DEV), torch.randn(4, device=DEV)
    for _ in range(3):
        pol.forward(obs); pol.backward(dm, dv, dl)
    read()
    n = 20
    for _ in range(n):
        pol.forward(obs)
    p = read()
    tot = sum(p[k] for k in FWD)
    print(f"forward  M={M}: block 0 total {tot / n:.0f} clk per launch")
    for k, name in FWD.items():
        print(f"   {name:28s} {p[k] / n:9.0f} clk  {100 * p[k] / tot:5.1f} %")
    for _ in range(n):
        pol.backward(dm, dv, dl)
    p = read()
    tot = sum(p[k] for k in BWD)
    print(f"backward M={M}: block 0 total {tot / n:.0f} clk per launch")
    for k, name in BWD.items():
        print(f"   {name:28s} {p[k] / n:9.0f} clk  {100 * p[k] / tot:5.1f} %")
