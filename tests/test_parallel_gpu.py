"""The N > 1 trainer path on real kernels: two processes share cuda:0 and exchange over gloo (the single-GPU test box
cannot host two RCCL ranks).  Checks what the 8-GPU run relies on: identical parameters on every rank after an update,
gradient = sum of the per-rank gradients that are already divided by the GLOBAL batch, per-minibatch advantage
statistics all-reduced."""
import os

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, algo):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from visfly_amd import parallel
    from _golden import ENV_DYN
    r, w, _ = parallel.init("gloo")
    dev = "cuda:0"
    if algo == "ppo":
        from visfly_amd.envs import HoverEnv
        from visfly_amd.ppo import PPO
        env = HoverEnv(num_agent_per_scene=1024, seed=10 + rank, dynamics_kwargs=dict(ENV_DYN), device=dev, max_episode_steps=64,
                       tensor_output=True)
        tr = PPO(env, n_steps=16, batch_size=4096, n_epochs=2, learning_rate=3e-4, seed=3)
        tr.learn(16 * 1024 * world * 2)
    else:
        from visfly_amd.bptt import BPTT
        from visfly_amd.envs import HoverEnv
        env = HoverEnv(num_agent_per_scene=512, seed=10 + rank, dynamics_kwargs=dict(ENV_DYN), device=dev, max_episode_steps=64,
                       tensor_output=True)
        tr = BPTT(env, horizon=8, learning_rate=1e-3, seed=3)
        tr.learn(8 * 512 * world * 3)
    flat = tr.policy.flat.detach().cpu()
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    ok = bool(torch.isfinite(flat).all()) and all(torch.equal(both[0], b) for b in both)
    q.put((rank, ok, tr.num_timesteps))
    dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["ppo", "bptt"])
def test_two_ranks_stay_in_lockstep(algo):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200 + (0 if algo == "ppo" else 1)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, algo)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = sorted(q.get() for _ in range(2))
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] for r in res), "parameters diverged between the ranks"
    assert res[0][2] == res[1][2] and res[0][2] > 0          # num_timesteps counts the global batch on every rank


def test_bench_two_ranks_control_flow():
    """bench.py --gpus 2 exactly as the driver launches it (torch.distributed.run), with the exchange over gloo because the
    test box has one GPU: barrier + max-over-ranks timing, ONE JSON line from rank 0, whole-job value"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VISFLY_AMD_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29900 + os.getpid() % 90), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "200",
           "--warmup", "20", "--agents", "16384"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 200 and out["scaling"] == "weak" and out["value"] > 0
    assert abs(out["value"] - 2 * 16384 * 200 / (out["ms_per_step"] * 1e-3 * 200)) < 1e-3 * out["value"]
    assert "cpu_baseline" not in out                       # rank 0 times the CPU port at N = 1 only
