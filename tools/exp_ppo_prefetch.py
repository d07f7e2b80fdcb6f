"""does preparing epoch e + 1's shuffled copy (vf_gather_rows + vf_adv_normalize_segments) on a side stream, under
epoch e's optimiser steps, shorten PPO.train()?  The experiment lives here, not in the product: `train_prefetch` below is PPO.train's loop
with two buffer sets and a side stream (same permutation draws, same minibatches -- checked bit-identical first).  A/B on bench.py's PPO
workload (25 600 agents, n_steps 256, batch 25 600, 5 epochs).  Result: profiles/r05_side_streams.txt.
Run on the GPU box: python tools/exp_ppo_prefetch.py [iters] [only]   (`only`: just the prefetching loop, for rocprofv3 + tools/trace_overlap.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from visfly_amd.envs import NavigationEnv
from visfly_amd.ppo import PPO


def train_prefetch(self):
    total = self.n_steps * self.n_envs
    bs = min(self.batch_size, total)
    g = th.Generator(device=self.device)
    g.manual_seed(self.seed + 7919 * (self._opt_step + 1))
    buf = self.buf
    flat = {"actions": buf.actions.view(-1, 4), "old_lp": buf.log_probs.view(-1), "adv": buf.advantages.view(-1), "ret": buf.returns.view(-1)}
    flat.update({"obs:" + k: buf.obs[k].view(-1, buf.obs[k].shape[-1]) for k in self.obs_keys})
    stats = th.zeros((self.n_epochs, 16), device=self.device)
    side = self.__dict__.setdefault("_side", th.cuda.Stream(device=self.device))
    main = th.cuda.current_stream(self.device)
    ready = self._prepare_epoch(flat, th.randperm(total, device=self.device, generator=g), bs, 0)
    for e in range(self.n_epochs):
        shuf, done = ready, None
        if e + 1 < self.n_epochs:
            perm = th.randperm(total, device=self.device, generator=g)          # drawn on the main stream, in the plain loop's order
            side.wait_stream(main)                                              # the other buffer set was read by epoch e - 1
            with th.cuda.stream(side):
                ready = self._prepare_epoch(flat, perm, bs, (e + 1) & 1)
                done = th.cuda.Event()
                done.record(side)
        for s in range(0, total, bs):
            self._minibatch_update({k: v[s:min(s + bs, total)] for k, v in shuf.items()}, stats[e])
        if done is not None:
            main.wait_event(done)


iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = "cuda:0"
dyn = dict(action_type="bodyrate", ori_output_type="quaternion", dt=0.0025, ctrl_dt=0.02, integrator="euler", drag_random=0.0)


def make(n, **kw):
    env = NavigationEnv(num_agent_per_scene=n, seed=1, device=dev, max_episode_steps=256, tensor_output=True, dynamics_kwargs=dict(dyn))
    return PPO(env, learning_rate=1e-4, seed=0, policy_kwargs=dict(activation_fn="relu"), **kw)


if len(sys.argv) <= 2:
    flats = []
    for fn in (PPO.train, train_prefetch):
        p = make(1024, n_steps=16, batch_size=6000, n_epochs=3)
        for _ in range(2):
            p.collect_rollouts()
            fn(p)
        th.cuda.synchronize()
        flats.append(p.policy.flat.clone())
    print("prefetching loop bit-identical to PPO.train:", bool(th.equal(flats[0], flats[1])), flush=True)
    assert th.equal(flats[0], flats[1])

ppo = make(25600, n_steps=256, batch_size=25600, n_epochs=5)
ppo.collect_rollouts()
if len(sys.argv) > 2:
    for _ in range(3):
        train_prefetch(ppo)
    th.cuda.synchronize()
    sys.exit(0)
rows = []
for rep in range(3):
    for name, fn in (("plain", PPO.train), ("prefetch", train_prefetch)):
        fn(ppo)
        th.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn(ppo)
        th.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        rows.append((name, dt * 1e3))
        print(f"rep {rep}  {name:9} train() {dt * 1e3:8.3f} ms", flush=True)
for name in ("plain", "prefetch"):
    v = sorted(r[1] for r in rows if r[0] == name)
    print(f"{name:9} median {v[1]:8.3f} ms  min {v[0]:8.3f} ms")
