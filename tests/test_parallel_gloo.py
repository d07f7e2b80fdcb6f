"""world_size-2 gloo runs on CPU: agent sharding covers every agent exactly once, the gradient
exchange (sum of per-rank gradients that are already divided by the GLOBAL batch) reproduces the
single-process gradient, and the global advantage statistics come out of the two all-reduced sums."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from visfly_amd import parallel


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = parallel.init("gloo")
    assert (r, w) == (rank, world) and parallel.world_size() == world
    N = 1001
    first, count = parallel.shard(N, r, w)
    owned = torch.zeros(N)
    owned[first:first + count] = 1
    parallel.allreduce_sum_(owned)
    assert (owned == 1).all()                                   # every agent owned exactly once
    # a toy "policy": linear regression loss = mean over the GLOBAL batch
    g = torch.Generator().manual_seed(0)
    X, y, wgt = torch.randn(N, 5, generator=g), torch.randn(N, generator=g), torch.randn(5, generator=g)
    full = (2 * (X @ wgt - y)[:, None] * X).mean(0)             # single-process gradient
    Xs, ys = X[first:first + count], y[first:first + count]
    local = (2 * (Xs @ wgt - ys)[:, None] * Xs).sum(0) / N      # per-rank rows, GLOBAL 1/B
    parallel.allreduce_sum_(local)
    assert torch.allclose(local, full, rtol=1e-5, atol=1e-6)
    # global advantage normalisation from (sum, sum of squares)
    adv = torch.randn(N, generator=g, dtype=torch.float64)
    a = adv[first:first + count]
    sums = torch.stack([a.sum(), (a * a).sum()])
    parallel.allreduce_sum_(sums)
    mean = sums[0] / N
    std = ((sums[1] - sums[0] * mean) / (N - 1)).sqrt()
    assert torch.allclose(mean, adv.mean()) and torch.allclose(std, adv.std())
    assert parallel.max_over_ranks(float(rank)) == world - 1
    parallel.barrier()
    q.put(rank)
    dist.destroy_process_group()


def test_two_process_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(2)) == [0, 1]


def test_shard_edges():
    assert parallel.shard(10, 0, 1) == (0, 10)
    assert [parallel.shard(10, r, 4) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]
    assert sum(parallel.shard(262144, r, 8)[1] for r in range(8)) == 262144


def test_bench_plain_command_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver starts the N = 1 run): bench.py re-runs itself
    as 2 ranks under torch.distributed.run on 127.0.0.1 and rank 0 prints ONE line -- rendezvous only (--launch-check, gloo);
    the GPU variant of the same command is tests/test_parallel_gpu.py::test_bench_two_ranks_control_flow[plain]"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["VISFLY_AMD_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"launch_check": True, "n_gpus": 2}, r.stdout[-2000:]
