"""r02: K env steps as K launches (step_n) vs ONE launch (step_n(fused=True)), 65 536 agents"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from visfly_amd.envs import HoverEnv, NavigationEnv  # noqa: E402

kw = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
dev = torch.device("cuda:0")
for cls, N in ((HoverEnv, 65536), (HoverEnv, 32768), (HoverEnv, 1048576), (NavigationEnv, 65536)):
    env = cls(num_agent_per_scene=N, dynamics_kwargs=dict(kw), device=dev, tensor_output=True, max_episode_steps=256)
    env.reset()
    g = torch.Generator(device=dev).manual_seed(1)
    for K in (20, 256):
        if N > 500000 and K > 20:
            continue
        A = ((torch.rand((K, N, 4), device=dev, generator=g) * 2 - 1) * 0.02 + torch.tensor([-1 / 3, 0, 0, 0], device=dev)).clamp(-1, 1).contiguous()
        res = {}
        for name, kwargs in (("launch per step", {}), ("fused", {"fused": True})):
            for _ in range(3):
                env.step_n(A, **kwargs)
            ws = []
            for _ in range(7):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                env.step_n(A, **kwargs)
                torch.cuda.synchronize()
                ws.append((time.perf_counter() - t0) * 1e6 / K)
            res[name] = statistics.median(ws)
        print(f"{cls.__name__} N={N} K={K}: " + "  ".join(f"{k} {v:.2f} us/step ({N / v * 1e6:.3e} agent-steps/s)" for k, v in res.items()))
    env.close()
