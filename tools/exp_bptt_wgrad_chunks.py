"""BPTT's weight-gradient launch (30 % of an update: X / dZ of the whole horizon, 3.9 GB, streamed back from HBM) -- would it pay to form it per
chunk of a few steps right behind the reverse sweep, while that chunk's rows are still in the 256 MiB Infinity Cache (verdict r05, item 6)?
The question can be answered without splitting the reverse sweep: the same launch over k steps' rows, timed cold (right after a pass over
the whole horizon evicted them) and hot (a second time in a row), against the one launch over all 64 steps; a chunked update would pay
64 / k hot launches (+ their folds) instead of the one.
    python tools/exp_bptt_wgrad_chunks.py            (BASELINE configs[4] shard: RacingEnv, thrust, 16 384 agents, H = 64, reference actor)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd.bptt import BPTT
from visfly_amd.envs import RacingEnv
DEV, N, H = "cuda:0", 16384, 64
dyn = dict(action_type="thrust", ori_output_type="quaternion", dt=0.0025, ctrl_dt=0.02, integrator="euler", drag_random=0.0, ctrl_delay=True)
env = RacingEnv(num_agent_per_scene=N, seed=42, dynamics_kwargs=dyn, device=DEV, max_episode_steps=256, requires_grad=True, tensor_output=True)
algo = BPTT(env, horizon=H, gamma=0.99, learning_rate=1e-3, seed=0, policy="MultiInputPolicy")
algo.learn(H * N * 2)          # the slots hold a real horizon's activations / masked gradients
torch.cuda.synchronize()
pol = algo.policy
os.environ["VISFLY_AMD_WGRAD_SPLIT_ROWS"] = "0"          # the raw launches first; the shipped split (equal runs of <= 524 288 rows) at the end
d_mu, d_ls = torch.randn((H, N, 4), device=DEV) / N, torch.randn((H, N, 4), device=DEV) / N
w = int(pol.n_params)
ev = lambda: torch.cuda.Event(enable_timing=True)


def timed(fn, reps=1):
    a, b = ev(), ev()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


full = lambda: pol.weight_grad_slots(N, H, d_mu, accumulate=True, d_value_all=d_ls)
full(); full()
t_full = min(timed(full) for _ in range(5))
rows_b = 0
nblk, blk = pol._slot_blocks[N]
print(f"reference actor: {w} parameters, {H} x {N} rows; the one launch over the horizon: {t_full:8.1f} us = {2.0 * w * H * N / t_full / 1e6:6.1f} TF/s")
print(f"{'steps':>5s} {'rows':>8s} {'cold us':>9s} {'hot us':>9s} {'hot TF/s':>9s} {'64/k hot launches, us':>22s} {'vs one launch':>14s}")
for k in (1, 2, 4, 8, 16, 32):
    cold, hot = [], []
    for rep in range(4):
        lo = (7 * rep * k) % (H - k + 1)
        part = lambda: pol.weight_grad_slots(N, k, d_mu[lo:lo + k], accumulate=True, d_value_all=d_ls[lo:lo + k], lo=lo)
        full()                               # evicts the chunk's rows (3.9 GB through a 256 MiB cache)
        a, m, b = ev(), ev(), ev()           # back to back behind it: the first pass finds the rows in HBM, the second in the cache
        a.record()
        part()
        m.record()
        part()
        b.record()
        torch.cuda.synchronize()
        cold.append(a.elapsed_time(m) * 1e3)
        hot.append(m.elapsed_time(b) * 1e3)
    c, h = min(cold), min(hot)
    print(f"{k:5d} {k * N:8d} {c:9.1f} {h:9.1f} {2.0 * w * k * N / h / 1e6:9.1f} {H / k * h:22.1f} {H / k * h / t_full:14.2f}")
os.environ["VISFLY_AMD_WGRAD_SPLIT_ROWS"] = "524288"
full(); full()
print(f"shipped: the horizon in equal runs of <= 524 288 rows: {min(timed(full) for _ in range(5)):8.1f} us")
