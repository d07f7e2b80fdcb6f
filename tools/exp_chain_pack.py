"""The fused PPO step of a generated chain class with every forward tile live vs with the ReLU masks kept as bits (PACK), over the row
count: does the register saving (two waves per SIMD) pay once there is more than one wave per SIMD to run?
Run twice on the GPU box: VISFLY_AMD_JIT_FLAGS=-DVF_GEN_LIVE_TILES=0 python tools/exp_chain_pack.py   (PACK at every size)
                          python tools/exp_chain_pack.py                                             (live tiles up to 24)
(the two plugin builds have different cache keys; build them with the same variable set before the gpurun)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd import _lib
from visfly_amd.ppo import MlpPolicy

DEV = "cuda:0"
dims = {"state": 13, "target": 3}
shapes = {"pi[128,128] vf[32] (23 tiles)": ({"state": [128, 64], "target": [128, 64]}, [128, 128], [32]),
          "pi[64,64] vf[64,32] (21 tiles)": ({"state": [128, 64], "target": [128, 64]}, [64, 64], [64, 32])}
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from visfly_amd import _jit
    for ext, pi, vf in shapes.values():
        print(_jit.build(_jit.shape_of(dims, ext, pi, vf)))
    sys.exit(0)
print("flags:", os.environ.get("VISFLY_AMD_JIT_FLAGS", "(default)"))
for name, (ext, pi, vf) in shapes.items():
    pol = MlpPolicy(dims, ext, pi, vf, DEV, seed=5, log_std_init=-0.3)
    assert pol.chain_jit
    for B in (25600, 65536, 262144, 524288):
        g = torch.Generator(device=DEV).manual_seed(B)
        obs = {k: torch.randn((B, d), device=DEV, generator=g) for k, d in dims.items()}
        actions = torch.tanh(torch.randn((B, 4), device=DEV, generator=g)).contiguous()
        old_lp, adv, ret = (torch.randn(B, device=DEV, generator=g) for _ in range(3))
        scratch = torch.zeros(16 * max(1024, (B + 31) // 32) + 4096, device=DEV)
        stats = torch.zeros(16, device=DEV)
        cfg = _lib.PpoLossCfg(0.2, 0.01, 0.5, 1.0 / B, pol.grad.data_ptr() + 4 * pol.log_std_off, None)
        for _ in range(3):
            assert pol.ppo_update(obs, actions, old_lp, adv, ret, cfg, stats, scratch)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        n = 20
        ev[0].record()
        for _ in range(n):
            pol.ppo_update(obs, actions, old_lp, adv, ret, cfg, stats, scratch)
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) * 1e3 / n
        print(f"{name:34} rows {B:7d}  fused step + weight gradients {us:9.1f} us   {6.0 * pol.log_std_off * B / us / 1e6 / 157.3:6.3f} of peak", flush=True)
    del pol
