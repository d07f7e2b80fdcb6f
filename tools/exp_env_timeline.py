"""timeline of k_env_step at 65 536 agents: wall-clock (s_memrealtime, 100 MHz) stamps of every main wave -- kernel entry, loads issued,
loads arrived, interval done, stores issued, stores acknowledged -- over consecutive launches.  Needs a -DVF_ENV_TRACE build:
  python -c "from visfly_amd import _build; _build.build(force=True, extra_flags=['-DVF_ENV_TRACE'], out='/tmp/libvf_trace.so')"
  VF_ALT_LIB=/tmp/libvf_trace.so python tools/exp_env_timeline.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from visfly_amd import _build, _lib
_build.LIB = _lib.LIB = os.environ['VF_ALT_LIB']
from visfly_amd.envs import HoverEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L = 24                                   # traced launches
kw = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
env = HoverEnv(num_agent_per_scene=N, dynamics_kwargs=kw, device="cuda:0", tensor_output=True, max_episode_steps=256)
env.reset()
a = (torch.rand((N, 4), device="cuda") * 2 - 1) * 0.02 + torch.tensor([-1 / 3, 0, 0, 0], device="cuda")
env.time_steps(a, 300)
waves = N // 64
buf = torch.zeros((L * waves + 64, 16), dtype=torch.int64, device="cuda")
lib = _lib.lib()
lib.vf_debug_env_trace.argtypes = [C.c_void_p, C.c_uint]
assert lib.vf_debug_env_trace(buf.data_ptr(), buf.shape[0]) == 0
us = env.time_steps(a, L)
torch.cuda.synchronize()
t = buf.cpu().numpy()
t = t[t[:, 0] > 0]
print(f"N={N} launches={L} waves traced={len(t)}  HIP-event us per launch {us:.2f}")
# group into launches by entry time: sort by t0, split into L groups of `waves`
t = t[np.argsort(t[:, 0])]
t0 = t[0, 0]
ns = lambda x: (x - t0) * 10.0            # 100 MHz -> ns
names = ["entry", "vector loads issued", "constant lines arrived", "kernarg lines arrived", "loads arrived", "head of sub-step 0", "head of sub-step 1",
         "interval done", "reward computed", "outputs written", "state stores issued", "all stores issued", "stores acked"]
NS = len(names)
prev_end = None
for l in range(len(t) // waves):
    g = t[l * waves:(l + 1) * waves]
    first, last_end = g[:, 0].min(), g[:, NS - 1].max()
    line = f"launch {l:2d}: first entry {ns(first) / 1000:8.2f} us"
    if prev_end is not None:
        line += f"  gap after previous launch's last ack {(first - prev_end) * 10 / 1000:5.2f} us"
    line += f"  entry spread {(g[:, 0].max() - first) * 10 / 1000:5.2f}  span (first entry -> last ack) {(last_end - first) * 10 / 1000:5.2f} us"
    print(line)
    prev_end = last_end
# per-phase statistics over launches 4.. (steady state), relative to the launch's first entry
rows = []
for l in range(4, len(t) // waves):
    g = t[l * waves:(l + 1) * waves].astype(np.float64)
    base = g[:, 0].min()
    rows.append((g[:, :NS] - base) * 0.01)
r = np.concatenate(rows)
print("phase (us after the launch's first wave entry):      min   median     p90      max")
for k, nme in enumerate(names):
    print(f"  {nme:24s} {r[:, k].min():8.2f} {np.median(r[:, k]):8.2f} {np.percentile(r[:, k], 90):8.2f} {r[:, k].max():8.2f}")
d = np.diff(r, axis=1)
print("per-wave phase durations (us):                        min   median     p90      max")
for k in range(NS - 1):
    print(f"  {names[k]:>24s} -> {names[k + 1]:24s} {d[:, k].min():6.2f} {np.median(d[:, k]):8.2f} {np.percentile(d[:, k], 90):8.2f} {d[:, k].max():8.2f}")
# by XCC
g = t[4 * waves:5 * waves]
x = (g[:, 14] >> 32) & 0xf
print("launch 4, by XCC id: waves, median entry, median end (us after first entry)")
for xi in np.unique(x):
    m = x == xi
    print(f"   xcc {xi}: {m.sum():4d} waves  entry {np.median(g[m, 0] - g[:, 0].min()) * 0.01:6.2f}  end {np.median(g[m, NS - 1] - g[:, 0].min()) * 0.01:6.2f}")
