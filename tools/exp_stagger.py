"""r02: de-phasing experiments for the env-step launch at 65 536 agents (one wave per SIMD: load burst | arithmetic | store burst).
   (1) VISFLY_AMD_STAGGER=q in the environment: half of the workgroups start q x 64 cycles late inside ONE launch
       (the hook -- `if (g.stagger > 0 && (blockIdx.x & 8)) for (q..) s_sleep(1)` at the top of k_env_step -- was removed again
       after the measurement: 12.28 / 12.25 / 12.29 / 12.37 / 12.52 / 12.89 us for q = 0 / 8 / 16 / 24 / 32 / 48).
   (2) two half-size envs on two streams, the second stream offset by a device-side sleep."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from visfly_amd.envs import HoverEnv  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
mode = sys.argv[2] if len(sys.argv) > 2 else "single"
kw = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
dev = torch.device("cuda:0")


def make(n):
    e = HoverEnv(num_agent_per_scene=n, dynamics_kwargs=dict(kw), device=dev, tensor_output=True, max_episode_steps=256)
    e.reset()
    return e


def actions(n, k):
    g = torch.Generator(device=dev).manual_seed(1)
    return ((torch.rand((k, n, 4), device=dev, generator=g) * 2 - 1) * 0.02 + torch.tensor([-1 / 3, 0, 0, 0], device=dev)).clamp(-1, 1).contiguous()


K = 400
if mode == "single":
    env = make(N)
    A = actions(N, 16).repeat((K // 16, 1, 1)).contiguous()
    print(f"stagger={os.environ.get('VISFLY_AMD_STAGGER', '0')} split={os.environ.get('VISFLY_AMD_SPLIT', 'auto')}: "
          f"kernel {env.time_steps(A[0], 400):.2f} us; ", end="")
    ws = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.step_n(A)
        torch.cuda.synchronize()
        ws.append((time.perf_counter() - t0) * 1e6 / K)
    print(f"step_n K={K}: {statistics.median(ws):.2f} us/step")
else:
    n = N // 2
    envs = [make(n), make(n)]
    acts = [actions(n, 16).repeat((K // 16, 1, 1)).contiguous() for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for off_us in (0, 2, 4, 5, 6, 8):
        def run():
            with torch.cuda.stream(streams[0]):
                envs[0].step_n(acts[0])
            with torch.cuda.stream(streams[1]):
                if off_us:
                    torch.cuda._sleep(int(off_us * 2100))      # ~2.1 GHz shader clock
                envs[1].step_n(acts[1])
        for _ in range(2):
            run()
        ws = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run()
            torch.cuda.synchronize()
            ws.append((time.perf_counter() - t0) * 1e6 / K)
        print(f"2 streams x {n} agents, stream 1 offset {off_us} us (split={os.environ.get('VISFLY_AMD_SPLIT', 'auto')}): "
              f"{statistics.median(ws):.2f} us per full step")
