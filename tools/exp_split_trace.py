"""placement and timeline of the waves of k_ppo_update_split (a -DVF_SPLIT_TRACE build as VF_ALT_LIB): which XCD / CU / SIMD every
wave ran on, waves per SIMD, wave life and phase lengths.   python tools/exp_split_trace.py [M]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visfly_amd import _build, _lib
if os.environ.get("VF_ALT_LIB"):
    _build.LIB = _lib.LIB = os.environ["VF_ALT_LIB"]
import numpy as np
import torch
from visfly_amd.ppo import MlpPolicy
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
DEV = "cuda:0"
pol = MlpPolicy({"state": 13, "target": 3}, {"state": [128, 64], "target": [128, 64]}, [64, 64], [64, 64], DEV, seed=9)
pol.lazy_pack = True
obs = {"state": torch.randn((M, 13), device=DEV), "target": torch.randn((M, 3), device=DEV)}
actions = torch.tanh(torch.randn((M, 4), device=DEV)).contiguous()
old_lp, adv, ret = torch.randn(M, device=DEV), torch.randn(M, device=DEV), torch.randn(M, device=DEV)
stats, scratch = torch.zeros(16, device=DEV), torch.zeros(16 * 1024, device=DEV)
cfg = _lib.PpoLossCfg(0.2, 0.0, 0.5, 1.0 / M, pol.grad.data_ptr() + 4 * pol.log_std_off, None)
for _ in range(20):
    assert pol.ppo_update(obs, actions, old_lp, adv, ret, cfg, stats, scratch)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB)
nw = 2 * ((M + 31) // 32)
buf = np.zeros((nw, 32), dtype=np.uint64)
rc = lib.vf_debug_split_trace(buf.ctypes.data_as(C.c_void_p), nw)
assert rc == 0, rc
hw, xcc = buf[:, 0].astype(np.int64) & 0xFFFFFFFF, (buf[:, 0].astype(np.int64) >> 32) & 0xF
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
skey = key * 4 + simd
t = buf[:, 3:8].astype(np.int64)
t0 = t[:, 0].min()
life = t[:, 4] - t[:, 0]
print(f"M={M}: {nw} waves, {len(np.unique(key))} CUs used, {len(np.unique(skey))} SIMDs used; waves per CU histogram {np.bincount(np.unique(key, return_counts=True)[1]).tolist()[:12]}; "
      f"waves per SIMD histogram {np.bincount(np.unique(skey, return_counts=True)[1]).tolist()[:6]}")
rt0, rt1 = buf[:, 2].astype(np.int64), buf[:, 1].astype(np.int64)       # 100 MHz, chip-wide
k0 = rt0.min()
print(f"   wave life mean {life.mean():.0f} min {life.min()} max {life.max()} cycles; effective clock {np.median(life / np.maximum(1, (rt1 - rt0)) / 10e-3 / 1e3):.2f} GHz")
print(f"   launch span (first entry -> last exit) {(rt1.max() - k0) / 100:.2f} us; entries spread over {(rt0.max() - k0) / 100:.2f} us (mean {np.mean(rt0 - k0) / 100:.2f}); "
      f"exits: p10 {np.percentile(rt1 - k0, 10) / 100:.2f} p50 {np.percentile(rt1 - k0, 50) / 100:.2f} p90 {np.percentile(rt1 - k0, 90) / 100:.2f} max {(rt1.max() - k0) / 100:.2f} us")
late = np.argsort(rt1)[-8:]
for w in late:
    print(f"      late wave {w} (role {w % 2}): entry {(rt0[w] - k0) / 100:.2f} us life {life[w]} cycles, SIMD holds {np.sum(skey == skey[w])}, CU holds {np.sum(key == key[w])} waves")
ph = np.diff(t, axis=1)
for r in (0, 1):
    sel = np.arange(nw) % 2 == r
    print(f"   role {r}: forward {ph[sel, 0].mean():.0f}  loss {ph[sel, 1].mean():.0f}  reverse {ph[sel, 2].mean():.0f}  tail {ph[sel, 3].mean():.0f} cycles")
# wave life by how many waves shared its SIMD
u, inv, cnt = np.unique(skey, return_inverse=True, return_counts=True)
for c in np.unique(cnt):
    sel = cnt[inv] == c
    print(f"   waves on SIMDs holding {c}: n={sel.sum()} life mean {life[sel].mean():.0f}")
# same-WG partner on the same SIMD?
same = (skey[0::2] == skey[1::2]).mean()
print(f"   partner waves on the same SIMD: {same * 100:.1f} %, on the same CU: {(key[0::2] == key[1::2]).mean() * 100:.1f} %")
# per-layer / per-op segments (cycle stamps at the end of every forward layer, before / after its epilogue, and of every reverse op)
ly = buf[:, 8:28].astype(np.int64)
for r in (0, 1):
    sel = np.arange(nw) % 2 == r
    pts = [("entry", t[sel, 0])]
    for li in range(5):
        pts.append((f"F{li} items", ly[sel, 0 + li]))
        pts.append((f"F{li} epi+xch", ly[sel, 5 + li]))
    pts.append(("loss", t[sel, 2]))
    for oi in range(4):
        pts.append((f"R{oi} items", ly[sel, 10 + oi]))
        pts.append((f"R{oi} fin+xch", ly[sel, 15 + oi]))
    pts.append(("exit", t[sel, 4]))
    segs = "  ".join(f"{pts[i][0]} {np.mean(pts[i][1] - pts[i - 1][1]):.0f}" for i in range(1, len(pts)))
    print(f"   role {r} segments (mean cycles): {segs}")
