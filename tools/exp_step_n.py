"""r02: host cost vs device time of the three ways to drive K env steps (65 536 agents by default):
   step() per call (fresh tensors / output ring), step_n() (launch loop in C), step_n(graph=True) (hipGraph replay);
   and the two-stream experiment: two half-size envs stepped on two streams so that one half's load / store bursts
   overlap the other half's arithmetic."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from visfly_amd.envs import HoverEnv  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
kw = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
dev = torch.device("cuda:0")


def make(n, **extra):
    e = HoverEnv(num_agent_per_scene=n, dynamics_kwargs=dict(kw), device=dev, tensor_output=True, max_episode_steps=256, **extra)
    e.reset()
    return e


def actions(n, k):
    g = torch.Generator(device=dev).manual_seed(1)
    return ((torch.rand((k, n, 4), device=dev, generator=g) * 2 - 1) * 0.02
            + torch.tensor([-1 / 3, 0, 0, 0], device=dev)).clamp(-1, 1).contiguous()


def region(fn, reps=9):
    """fn enqueues K steps; returns (median wall us/step between synchronizes, median host-enqueue us/step)"""
    walls, hosts = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        walls.append((t2 - t0) * 1e6 / K)
        hosts.append((t1 - t0) * 1e6 / K)
    return statistics.median(walls), statistics.median(hosts), min(walls)


A = actions(N, K)
env = make(N)
ring = make(N, out_buffers=4)
kern = env.time_steps(A[0], 300)
print(f"N={N} K={K}: kernel {kern:.2f} us (HIP events, 300 back-to-back launches)")
for name, fn in (
        ("step() fresh tensors", lambda: [env.step(A[k]) for k in range(K)]),
        ("step() out_buffers=4", lambda: [ring.step(A[k]) for k in range(K)]),
        ("step_n() C loop", lambda: env.step_n(A)),
        ("step_n(graph=True)", lambda: env.step_n(A, graph=True)),
):
    for _ in range(3):
        fn()
    w, h, mn = region(fn)
    print(f"  {name:24s} wall {w:7.2f} us/step (min {mn:6.2f})   host enqueue {h:6.2f} us/step   wall/kernel {w / kern:.2f}")

# long regions (host cost amortised): K = 400
K_long = 400
AL = actions(N, 16)
AL = AL.repeat((K_long // 16, 1, 1)).contiguous()
for _ in range(2):
    env.step_n(AL)
torch.cuda.synchronize()
t0 = time.perf_counter()
env.step_n(AL)
torch.cuda.synchronize()
print(f"  step_n K={K_long}: {(time.perf_counter() - t0) * 1e6 / K_long:.2f} us/step")

# ---- two streams, two half-size envs (plain kernel forced by VISFLY_AMD_SPLIT=0 in the environment if wanted)
for parts in (2, 4):
    n = N // parts
    envs = [make(n) for _ in range(parts)]
    acts = [actions(n, K_long) for _ in range(parts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]

    def run():
        for e, a, s in zip(envs, acts, streams):
            with torch.cuda.stream(s):
                e.step_n(a)

    for _ in range(2):
        run()
    torch.cuda.synchronize()
    ws = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        ws.append((time.perf_counter() - t0) * 1e6 / K_long)
    one = envs[0].time_steps(acts[0][0], 300)
    print(f"  {parts} streams x {n} agents: {statistics.median(ws):.2f} us per full step (all {N} agents); "
          f"a lone {n}-agent launch takes {one:.2f} us")
