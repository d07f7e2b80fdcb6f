"""a few launches of one linear forward shape (for rocprofv3 --pmc passes)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd import _lib
lib = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
M, K, No = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
X = torch.randn((M, K), device="cuda"); W = torch.randn((No, K), device="cuda"); b = torch.randn(No, device="cuda")
Y = torch.empty((M, No), device="cuda")
for _ in range(10):
    lib.vf_linear_fwd(X.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), No, M, K, No, 1, st)
torch.cuda.synchronize()
