"""Does the env step's duration depend on WHERE its per-step buffers lie?  (round 2: bench.py --chunk 4 gave 12.0 us per
step, --chunk 16 10.7 us, reproducibly, and the only difference is what torch's allocator handed out.)

One arena; the (K,N,..) action / obs / reward / done buffers of step_n are placed at chosen byte offsets inside it and the
same 1000 steps are timed for each placement.  Output: gpurun_out/placement.csv (offsets, slab address, us per step).

    python tools/exp_placement.py [N] [K]
"""
import ctypes as C
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch as th

from visfly_amd import _lib
from visfly_amd.envs import HoverEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
STEPS = 1024
dev = th.device("cuda", 0)
from bench import DYN_KW
env = HoverEnv(num_agent_per_scene=N, num_scene=1, seed=42, visual=False, dynamics_kwargs=DYN_KW, device=dev,
               max_episode_steps=256, tensor_output=True)
env.reset()
ARENA = 1 << 30
arena = th.empty(ARENA, dtype=th.uint8, device=dev)
base = arena.data_ptr()
slab = env._slab.data_ptr()
sizes = dict(act=K * N * 16, obs=K * N * 52, rew=K * N * 4, done=K * N)
g = th.Generator(device=dev).manual_seed(0)
hover = th.tensor([-1 / 3, 0, 0, 0], device=dev)
pool = (hover + (th.rand((K, N, 4), device=dev, generator=g) * 2 - 1) * 0.02).clamp(-1, 1).contiguous()


def view(off, nbytes, dtype, shape):
    return arena[off:off + nbytes].view(dtype).view(shape)


def measure(offs):
    act = view(offs["act"], sizes["act"], th.float32, (K, N, 4))
    act.copy_(pool)
    obs = view(offs["obs"], sizes["obs"], th.float32, (K, N, 13))
    rew = view(offs["rew"], sizes["rew"], th.float32, (K, N))
    done = view(offs["done"], sizes["done"], th.bool, (K, N))
    r = _lib.EnvRollout()
    r.out = env._out(obs, rew, done)
    r.action_stride, r.obs_stride, r.reward_stride, r.done_stride = 4 * N, 13 * N, N, N
    r.K = K
    env._rollouts.clear()
    env._rollouts[K] = {"obs": obs, "reward": rew, "done": done, "r": r, "ref": C.byref(r), "graphs": {}}
    for _ in range(4):
        env.step_n(act)
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(STEPS // K):
            env.step_n(act)
        e1.record()
        th.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (STEPS // K * K))
    return best


def pack(order, pad, start):
    offs, cur = {}, start
    for k in order:
        offs[k] = cur
        cur = (cur + sizes[k] + pad + 255) // 256 * 256
    return offs


rows = []
os.makedirs("gpurun_out", exist_ok=True)
# 1. packed back to back, the group shifted as a whole
for start in [0, 256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 16 << 20, 17 << 20, 64 << 20, 100 << 20]:
    offs = pack(["act", "obs", "rew", "done"], 0, start)
    rows.append(("shift", start, 0, offs, measure(offs)))
# 2. padding between the buffers
for pad in [256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, 2 << 20, (2 << 20) + 4096, 8 << 20]:
    offs = pack(["act", "obs", "rew", "done"], pad, 0)
    rows.append(("pad", 0, pad, offs, measure(offs)))
# 3. random placements (multiples of 256 B, non-overlapping by construction: four 256 MB quarters)
random.seed(1)
for i in range(120):
    offs = {}
    for q, k in enumerate(random.sample(["act", "obs", "rew", "done"], 4)):
        room = (ARENA // 4) - sizes[k]
        offs[k] = q * (ARENA // 4) + random.randrange(0, room // 256) * 256
    rows.append(("random", 0, 0, offs, measure(offs)))
with open("gpurun_out/placement.csv", "w") as f:
    f.write(f"# N={N} K={K} arena={base:#x} slab={slab:#x}\nkind,start,pad,act,obs,rew,done,us_per_step\n")
    for kind, start, pad, offs, us in rows:
        f.write(f"{kind},{start},{pad},{offs['act']},{offs['obs']},{offs['rew']},{offs['done']},{us:.4f}\n")
us = sorted(r[4] for r in rows)
print(f"N={N} K={K} arena={base:#x} slab={slab:#x}")
print(f"{len(rows)} placements: min {us[0]:.3f}  p10 {us[len(us) // 10]:.3f}  median {us[len(us) // 2]:.3f}  p90 {us[len(us) * 9 // 10]:.3f}  max {us[-1]:.3f} us per step")
for kind in ("shift", "pad"):
    print(kind, " ".join(f"{(r[1] if kind == 'shift' else r[2])}:{r[4]:.2f}" for r in rows if r[0] == kind))
