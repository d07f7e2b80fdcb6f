"""Is k_mlp_wgrad's 147 MB of X / dZ served from the 256 MiB Infinity Cache or from HBM?  
FETCH_SIZE cannot tell (the L2's fabric-side counters see Infinity-Cache hits too, MI355X_MICROARCH.md), so time the launch
  (a) right behind the chain kernel that wrote its operands (the optimiser step's own order),
  (b) behind a 1 GiB streaming write + read that evicts everything the chain kernel left on the die,
  (c) a second time on the same operands (read-after-read),
HIP events on the launch stream around the weight-gradient launch alone (k_mlp_wgrad + k_wgrad_fold)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if len(sys.argv) > 3:            # A/B: another build of the library (python -c "from visfly_amd import _build; _build.build(force=True, extra_flags=[..], out=..)")
    from visfly_amd import _build
    _build.LIB = sys.argv[3]
from visfly_amd import _lib
if len(sys.argv) > 3:
    _lib.LIB = sys.argv[3]
from visfly_amd.ppo import MlpPolicy, _ptr

M = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 30
DEV = "cuda:0"
pol = MlpPolicy({"state": 13, "target": 3}, {"state": [128, 64], "target": [128, 64]}, [64, 64], [64, 64], DEV, seed=9)
pol.lazy_pack = True
obs = {"state": torch.randn((M, 13), device=DEV), "target": torch.randn((M, 3), device=DEV)}
actions = torch.tanh(torch.randn((M, 4), device=DEV)).contiguous()
old_lp, adv, ret = torch.randn(M, device=DEV), torch.randn(M, device=DEV), torch.randn(M, device=DEV)
stats, scratch = torch.zeros(16, device=DEV), torch.zeros(16 * 1024, device=DEV)
cfg = _lib.PpoLossCfg(0.2, 0.0, 0.5, 1.0 / M, pol.grad.data_ptr() + 4 * pol.log_std_off, None)
for _ in range(5):
    assert pol.ppo_update(obs, actions, old_lp, adv, ret, cfg, stats, scratch)
torch.cuda.synchronize()
L, st = _lib.lib(), pol._stream()
d = pol._descs[(M, 0, True)]
bd, firsts = pol._descs[("ppo_bwd", M)]
b = pol._buffers(M, 0)
ins = [_ptr(b["obs:" + k]) for k in pol.obs_keys]
big = torch.empty(1 << 28, dtype=torch.float32, device=DEV)      # 1 GiB


def chain():
    _lib.check(L.vf_ppo_update(C.byref(d), C.byref(bd), _ptr(pol.flat), _ptr(pol._packed), ins[0], ins[1], _ptr(pol.log_std),
                               _ptr(actions), _ptr(old_lp), _ptr(adv), _ptr(ret), _ptr(stats), M, C.byref(cfg), _ptr(scratch), st))


def wgrad():
    _lib.check(L.vf_mlp_weight_grad(C.byref(bd), _ptr(pol._scratch), _ptr(pol.grad), M, 0, st))


def timed(pre, n=REP):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in ev:
        pre()
        e0.record()
        wgrad()
        e1.record()
    torch.cuda.synchronize()
    t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    return t[len(t) // 2], t[0], t[-1]


def flush():
    big.fill_(1.0)
    big.sum()


def chain_flush():
    chain()
    flush()


print(f"weight-gradient launch (k_mlp_wgrad + k_wgrad_fold) at M = {M}, median / min / max us over {REP} runs (HIP events)")
print("  (a) right behind the chain kernel        %.1f / %.1f / %.1f" % timed(chain))
print("  (b) chain, then 1 GiB written and read   %.1f / %.1f / %.1f" % timed(chain_flush))
print("  (c) behind another weight-gradient run   %.1f / %.1f / %.1f" % timed(wgrad))
print("  (d) behind the 1 GiB flush alone         %.1f / %.1f / %.1f" % timed(flush))
print("  (a) again                                %.1f / %.1f / %.1f" % timed(chain))
