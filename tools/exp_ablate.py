"""ablation of the linear forward kernel: builds with -DVF_ABLATE=n (1: one MFMA chunk only, 2: no epilogue stores)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from visfly_amd import _build
code = r'''
import sys
sys.path.insert(0, %r)
import torch
from visfly_amd import _build, _lib
_lib.LIB = sys.argv[1]
lib = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
out = []
for M, K, No in ((25600, 128, 64), (25600, 64, 64), (64, 128, 64), (25600, 64, 4)):
    X = torch.randn((M, K), device="cuda"); W = torch.randn((No, K), device="cuda"); b = torch.randn(No, device="cuda")
    Y = torch.empty((M, No), device="cuda")
    out.append(round(timeit(lambda: lib.vf_linear_fwd(X.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), No, M, K, No, 1, st)), 1))
print(out)
''' % root
if __name__ == "__main__":
    for lib in sys.argv[1:]:
        r = subprocess.run([sys.executable, "-c", code, lib], capture_output=True, text=True)
        print(os.path.basename(lib), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
