"""The N > 1 trainer path on real kernels: two processes share cuda:0 and exchange over gloo (the single-GPU test box
cannot host two RCCL ranks).  Checks what the 8-GPU run relies on: identical parameters on every rank after an update,
gradient = sum of the per-rank gradients that are already divided by the GLOBAL batch, per-minibatch advantage
statistics all-reduced."""
import os

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, algo, backend="gloo", buckets=1):
    import sys
    os.environ["VISFLY_AMD_GRAD_BUCKETS"] = str(buckets)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    local = rank if backend == "nccl" else 0
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from visfly_amd import parallel
    from _golden import ENV_DYN
    r, w, _ = parallel.init(backend)
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    if algo == "ppo":
        from visfly_amd.envs import HoverEnv
        from visfly_amd.ppo import PPO
        env = HoverEnv(num_agent_per_scene=1024, seed=10 + rank, dynamics_kwargs=dict(ENV_DYN), device=dev, max_episode_steps=64,
                       tensor_output=True)
        tr = PPO(env, n_steps=16, batch_size=4096, n_epochs=2, learning_rate=3e-4, seed=3 + 17 * rank, policy_kwargs=dict(activation_fn="relu"))   # rank-dependent seed: the ctor broadcasts rank 0's weights
        tr.learn(16 * 1024 * world * 2)
    elif algo == "shac":
        from visfly_amd.envs import HoverEnv
        from visfly_amd.shac import SHAC
        env = HoverEnv(num_agent_per_scene=512, seed=10 + rank, dynamics_kwargs=dict(ENV_DYN), device=dev, max_episode_steps=64,
                       tensor_output=True, requires_grad=True)
        tr = SHAC(env, horizon=8, learning_rate=1e-3, gradient_steps=2, seed=3 + 17 * rank)
        tr.learn(8 * 512 * world * 3)
    else:
        from visfly_amd.bptt import BPTT
        from visfly_amd.envs import HoverEnv
        env = HoverEnv(num_agent_per_scene=512, seed=10 + rank, dynamics_kwargs=dict(ENV_DYN), device=dev, max_episode_steps=64,
                       tensor_output=True)
        tr = BPTT(env, horizon=8, learning_rate=1e-3, seed=3 + 17 * rank)
        tr.learn(8 * 512 * world * 3)
    flat = tr.policy.flat.detach()
    if algo == "shac":       # the critic (fused update step, its own all-reduce) must stay in lockstep too
        flat = torch.cat([flat, tr.critic.flat.detach(), tr.critic_target.flat.detach()])
    flat = flat if backend == "nccl" else flat.cpu()
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    ok = bool(torch.isfinite(flat).all()) and all(torch.equal(both[0], b) for b in both)
    if backend == "nccl":
        ok = ok and parallel.native_comm() is not None          # the gradient went through vf_allreduce_grads
    import hashlib
    digest = hashlib.sha256(flat.cpu().numpy().tobytes()).hexdigest()
    q.put((rank, ok, tr.num_timesteps, digest, getattr(tr, "grad_buckets", 1)))
    dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["ppo", "bptt", "shac"])
def test_two_ranks_stay_in_lockstep(algo):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200 + {"ppo": 0, "bptt": 1, "shac": 2}[algo]
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


def test_two_bucket_gradient_exchange_equals_one_bucket():
    """VISFLY_AMD_GRAD_BUCKETS=2 (trunks' gradients + log_std + statistics all-reduced on a second stream under the extractors'
    weight-gradient launches, the extractors' bucket behind it, Adam behind both) ends at the parameters of the one-bucket exchange, bit
    for bit: the same per-element sums, only issued as two collectives"""
    ctx = mp.get_context("spawn")
    digests = {}
    for buckets in (1, 2):
        q = ctx.Queue()
        port = 29850 + os.getpid() % 100 + buckets
        procs = [ctx.Process(target=_worker, args=(r, 2, port, q, "ppo", "gloo", buckets)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        res = sorted(q.get() for _ in range(2))
        assert all(r[1] for r in res) and res[0][3] == res[1][3] and all(r[4] == buckets for r in res), res
        digests[buckets] = res[0][3]
    assert digests[1] == digests[2]


@pytest.mark.parametrize("algo", ["ppo", "bptt"])
def test_two_ranks_rccl(algo):
    """the same lock-step check over RCCL (backend "nccl", gradient through vf_allreduce_grads) -- needs two GPUs; the
    single-GPU test box skips it, an 8-GPU node runs it"""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29750 + os.getpid() % 100 + (0 if algo == "ppo" else 1)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, algo, "nccl", 2 if algo == "ppo" else 1)) for r in range(2)]      # (PPO: the two-bucket exchange over RCCL)
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = sorted(q.get() for _ in range(2))
    assert all(r[1] for r in res), "parameters diverged between the ranks (or the native communicator was not used)"


def test_native_rccl_communicator_single_rank():
    """vf_comm_* / vf_allreduce_grads on a world-of-one communicator: RCCL resolved from the library torch already maps,
    the unique id / init / all-reduce / destroy sequence runs on this process's stream; a sum over one rank is the identity"""
    import ctypes as C
    from visfly_amd import _lib, parallel
    L = _lib.lib()
    assert L.vf_allreduce_grads(None, None, 1, None) == -1
    c = parallel.native_comm(force=True)
    assert c is not None, "RCCL communicator could not be created"
    g = torch.randn(43977 + 16, device="cuda:0")
    g0 = g.clone()
    _lib.check(L.vf_allreduce_grads(c, g.data_ptr(), g.numel(), _lib.current_stream(g.device)))
    d = torch.randn(14, dtype=torch.float64, device="cuda:0")
    d0 = d.clone()
    _lib.check(L.vf_allreduce_f64(c, d.data_ptr(), d.numel(), _lib.current_stream(d.device)))
    torch.cuda.synchronize()
    assert torch.equal(g, g0) and torch.equal(d, d0)
    assert L.vf_allreduce_grads(c, None, 4, None) == -1 and L.vf_allreduce_grads(c, g.data_ptr(), 0, None) == -1
    ident = (C.c_uint8 * 128)()
    _lib.check(L.vf_comm_unique_id(ident))
    assert any(ident)
    h = _lib._vp()
    assert L.vf_comm_init(ident, 2, 5, C.byref(h)) == -1          # rank outside the world


@pytest.mark.parametrize("launcher", ["torchrun", "plain"])
def test_bench_two_ranks_control_flow(launcher):
    """bench.py --gpus 2 both ways the driver may start it -- under torch.distributed.run, and as the plain
    `python bench.py --gpus 2` (bench.py then spawns its own ranks, bench._self_launch) -- with the exchange over gloo because
    the test box has one GPU: barrier + max-over-ranks timing, ONE JSON line from rank 0, whole-job value"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VISFLY_AMD_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "200", "--warmup", "20", "--agents", "16384"]
    if launcher == "torchrun":     # (the primary line only: the secondary legs' two-rank control flow is the plain launch's below)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(29900 + os.getpid() % 90)] + tail + ["--no-secondary"]
    else:
        cmd = [sys.executable] + tail
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 200 and out["scaling"] == "weak" and out["value"] > 0
    assert abs(out["value"] - 2 * 16384 * 200 / (out["ms_per_step"] * 1e-3 * 200)) < 1e-3 * out["value"]
    assert "cpu_baseline" not in out                       # rank 0 times the CPU port at N = 1 only
