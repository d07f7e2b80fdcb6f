"""microbenchmark of the MFMA linear kernels per shape (torch.cuda events, current stream)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd import _lib
lib = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for M in (25600, 32768):
    for K, No in ((13, 128), (128, 64), (64, 64), (64, 4), (64, 1), (128, 128)):
        X = torch.randn((M, K), device="cuda"); W = torch.randn((No, K), device="cuda"); b = torch.randn(No, device="cuda")
        Y = torch.empty((M, No), device="cuda"); dY = torch.randn((M, No), device="cuda"); dX = torch.empty((M, K), device="cuda")
        dW = torch.empty((No, K), device="cuda"); db = torch.empty(No, device="cuda")
        scr = torch.empty(int(lib.vf_linear_bwd_scratch_floats(M, K, No)), device="cuda")
        f = timeit(lambda: lib.vf_linear_fwd(X.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), No, M, K, No, 1, st))
        d = timeit(lambda: lib.vf_linear_bwd_data(dY.data_ptr(), No, Y.data_ptr(), No, W.data_ptr(), dX.data_ptr(), K, M, K, No, 0, st))
        w = timeit(lambda: lib.vf_linear_bwd_weight(dY.data_ptr(), No, Y.data_ptr(), No, X.data_ptr(), K, dW.data_ptr(), db.data_ptr(), M, K, No, scr.data_ptr(), st))
        t = timeit(lambda: torch.relu(torch.addmm(b, X, W.T)))
        gf = 2 * M * K * No / 1e3
        print(f"M={M} K={K:3d} No={No:3d}: fwd {f:6.1f} us ({gf/f/1e3:6.2f} TF/s)  bwd_data {d:6.1f}  bwd_weight {w:6.1f}  | torch addmm+relu {t:6.1f}")
