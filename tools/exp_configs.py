"""fused env-step launch time (HIP events) for the BASELINE.json configurations on one GPU"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get('VF_ALT_LIB'):     # another build of the library (A/B)
    from visfly_amd import _build, _lib
    _build.LIB = _lib.LIB = os.environ['VF_ALT_LIB']
from visfly_amd.envs import HoverEnv, NavigationEnv, RacingEnv
DEV = "cuda:0"
base = dict(dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0., 2., 1.]}}]}}
cases = [
    ("configs[0] HoverEnv bodyrate euler", HoverEnv, 128, dict(base, action_type="bodyrate", integrator="euler"), {}, [-1 / 3, 0, 0, 0]),
    ("configs[1] HoverEnv bodyrate euler", HoverEnv, 65536, dict(base, action_type="bodyrate", integrator="euler"), {}, [-1 / 3, 0, 0, 0]),
    ("configs[2] NavigationEnv bodyrate rk4 drag_random", NavigationEnv, 65536, dict(base, action_type="bodyrate", integrator="rk4", drag_random=0.1),
     dict(random_kwargs=spawn), [-1 / 3, 0, 0, 0]),
    ("configs[3] NavigationEnv bodyrate euler (PPO shard)", NavigationEnv, 32768, dict(base, action_type="bodyrate", integrator="euler"),
     dict(random_kwargs=spawn), [-1 / 3, 0, 0, 0]),
    ("configs[4] RacingEnv thrust euler (BPTT shard)", RacingEnv, 16384, dict(base, action_type="thrust", integrator="euler"), {}, [-0.8333] * 4),
    ("f1 HoverEnv velocity euler", HoverEnv, 65536, dict(base, action_type="velocity", integrator="euler"), {}, [0, 0, 0, 0]),
    ("f1 HoverEnv position euler", HoverEnv, 65536, dict(base, action_type="position", integrator="euler"), {}, [0, 0.1, 0, 0.15]),
    ("configs[3] shape, transcendentals=sleef", NavigationEnv, 32768, dict(base, action_type="bodyrate", integrator="euler", transcendentals="sleef"),
     dict(random_kwargs=spawn), [-1 / 3, 0, 0, 0]),
    ("f1 velocity, transcendentals=sleef", HoverEnv, 65536, dict(base, action_type="velocity", integrator="euler", transcendentals="sleef"), {}, [0, 0, 0, 0]),
    ("1M agents HoverEnv bodyrate euler", HoverEnv, 1 << 20, dict(base, action_type="bodyrate", integrator="euler"), {}, [-1 / 3, 0, 0, 0]),
]
for name, cls, N, dkw, kw, hover in cases:
    env = cls(num_agent_per_scene=N, seed=1, dynamics_kwargs=dkw, device=DEV, max_episode_steps=256, tensor_output=True, **kw)
    env.reset()
    a = (torch.tensor(hover, device=DEV) + (torch.rand((N, 4), device=DEV) * 2 - 1) * 0.02).clamp(-1, 1).contiguous()
    us = min(env.time_steps(a, iters=200) for _ in range(3))
    print(f"{name:55s} N={N:8d}  {us:8.2f} us/step  {N / us * 1e6:.3e} agent-steps/s")
    del env
