"""a few fused PPO minibatch steps (vf_ppo_update + vf_mlp_weight_grad) of the Nav actor-critic at M rows, for rocprofv3 passes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visfly_amd import _build, _lib
if os.environ.get('VF_ALT_LIB'):     # another build of the library (A/B)
    _build.LIB = _lib.LIB = os.environ['VF_ALT_LIB']
import torch
from visfly_amd.ppo import MlpPolicy
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
DEV = "cuda:0"
pol = MlpPolicy({"state": 13, "target": 3}, {"state": [128, 64], "target": [128, 64]}, [64, 64], [64, 64], DEV, seed=9)
pol.lazy_pack = True
obs = {"state": torch.randn((M, 13), device=DEV), "target": torch.randn((M, 3), device=DEV)}
actions = torch.tanh(torch.randn((M, 4), device=DEV)).contiguous()
old_lp, adv, ret = torch.randn(M, device=DEV), torch.randn(M, device=DEV), torch.randn(M, device=DEV)
stats, scratch = torch.zeros(16, device=DEV), torch.zeros(16 * 1024, device=DEV)
cfg = _lib.PpoLossCfg(0.2, 0.0, 0.5, 1.0 / M, pol.grad.data_ptr() + 4 * pol.log_std_off, None)
for _ in range(n):
    assert pol.ppo_update(obs, actions, old_lp, adv, ret, cfg, stats, scratch)
torch.cuda.synchronize()
