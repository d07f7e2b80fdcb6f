"""Wave timeline of the fused optimiser tail (k_mlp_wgrad<false, true>): a -DVF_WGRAD_TRACE build leaves the 100 MHz clock of every wave at
its phase boundaries; this prints, per phase, when the first / median / last wave passes it, relative to the first wave's start.
    python tools/build_variant.py vf_mlp_wgrad.hip tools/tmp/libvf_wtrace.so -DVF_WGRAD_TRACE
    VF_ALT_LIB=tools/tmp/libvf_wtrace.so python tools/exp_tail_trace.py [rows]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visfly_amd import _build, _lib
if os.environ.get("VF_ALT_LIB"):
    _build.LIB = _lib.LIB = os.path.abspath(os.environ["VF_ALT_LIB"])
    _build.build = lambda *a, **k: _lib.LIB
import numpy as np, torch
from visfly_amd.ppo import MlpPolicy
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
trace = torch.zeros(8 * 2048, dtype=torch.int64, device=DEV)
os.environ["VISFLY_AMD_WGRAD_TRACE_PTR"] = str(trace.data_ptr())
L = _lib.lib()
dims = {"state": 13, "target": 3}
g = torch.Generator(device=DEV).manual_seed(0)
obs = {k: torch.randn((B, dims[k]), device=DEV, generator=g) for k in dims}
actions = torch.tanh(torch.randn((B, 4), device=DEV, generator=g)).contiguous()
old_lp, adv, ret = (torch.randn(B, device=DEV, generator=g) for _ in range(3))
pol = MlpPolicy(dims, {k: [128, 64] for k in dims}, [64, 64], [64, 64], DEV, seed=5)
pol.lazy_pack = True
n = pol.n_params
gbuf = torch.zeros(n + 16, device=DEV)
pol.grad = gbuf[:n]
stats, acc, scratch = gbuf[n:], torch.zeros(16, device=DEV), torch.zeros(16 * 1024 + 4096, device=DEV)
m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
sync = torch.zeros(_lib.WGRAD_SYNC_WORDS, dtype=torch.int32, device=DEV)
names = ["start", "slab done", "layer met", "block folded", "all met + norm", "(unused)", "end"]
allt = []
for step in range(1, 41):
    cfg = _lib.PpoLossCfg(0.2, 0.0, 0.5, 1.0 / B, pol.grad.data_ptr() + 4 * pol.log_std_off, acc.data_ptr())
    pmap, packed = pol.pack_map()
    acfg = _lib.AdamCfg(1e-4, 0.9, 0.999, 1e-8, 1e-5, 0.5, step, 0, pmap.data_ptr(), packed.data_ptr(), None, 0, pol.log_std_off)
    tail = _lib.WgradTail(pol.flat.data_ptr(), m.data_ptr(), v.data_ptr(), n, acfg, sync.data_ptr())
    trace.zero_()
    res = pol.ppo_update(obs, actions, old_lp, adv, ret, cfg, stats, scratch, want_sumsq=True, tail=tail)
    assert res == "adam", pol.tail_reason
    pol.mark_updated(packed_current=True)
    torch.cuda.synchronize()
    t = trace.view(-1, 8).cpu().numpy()
    t = t[t[:, 0] > 0][:, :7].astype(np.float64)
    if step > 20:
        allt.append((t - t[:, 0].min()) / 100.0)        # us
waves = allt[0].shape[0]
T = np.stack(allt)           # (launches, waves, 7)
print(f"{B} rows, {waves} waves, {len(allt)} launches; us since the first wave's start: first / median / last wave (mean over launches)")
for k, nm in enumerate(names):
    c = T[:, :, k]
    print(f"  {nm:14s} {c.min(1).mean():8.2f} {np.median(c, 1).mean():8.2f} {c.max(1).mean():8.2f}")
d = np.diff(T, axis=2)
print("phase durations per wave (us): median / max over waves (mean over launches)")
for k in range(6):
    print(f"  {names[k]:>13s} -> {names[k + 1]:14s} {np.median(d[:, :, k], 1).mean():8.2f} {d[:, :, k].max(1).mean():8.2f}")

# per layer: how long the slab loop took (the plan hands out rows in proportion to tiles + 2 per row pair: is that the cost?)
bd = pol._descs[("ppo_bwd", B)][0]
tiles = [((bd.layer[l].No + 31) // 32) * ((bd.layer[l].K + 31) // 32) + 1 for l in range(bd.n_layers)]
budget, first = 1024, [0]
while True:
    first, rows_l = [0], []
    for l in range(bd.n_layers):
        nw = max(1, (budget * tiles[l] + sum(tiles) // 2) // sum(tiles))
        rows = max(2, ((B + nw - 1) // nw + 1) & ~1)
        first.append(first[-1] + (B + rows - 1) // rows)
        rows_l.append(rows)
    if first[-1] <= 1024 or budget <= 8:
        break
    budget -= 8
assert first[-1] == waves, (first, waves)
slab = T[:, :, 1] - T[:, :, 0]
print("per layer (No x K): waves, rows per wave, slab loop us median / max, us per (row pair x tile-units)")
for l in range(bd.n_layers):
    c = slab[:, first[l]:first[l + 1]]
    print(f"  layer {l:2d} {bd.layer[l].No:4d} x {bd.layer[l].K:4d}  waves {first[l + 1] - first[l]:4d} rows {rows_l[l]:5d}  {np.median(c):7.2f} {c.max(1).mean():7.2f}"
          f"   {np.median(c) / (rows_l[l] / 2) * 1e3:7.2f} ns per row pair ({tiles[l] - 1} tiles)")
