"""long-run soak of the fused env step: every env kind, 65 536 agents, U(-1,1) actions (crashes / re-spawns in every step),
`steps` control steps through step_n chunks; every output and the slab must stay finite, counters in range, done rate sane.
    python tools/soak_envs.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
import visfly_amd.envs as E

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N, K = 65536, 250
NAV_RK = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0.0, 2., 1.]}}]}}
CASES = [("HoverEnv", dict(action_type="bodyrate", integrator="euler"), {}),
         ("HoverEnv", dict(action_type="thrust", integrator="rk4"), {}),
         ("HoverEnv", dict(action_type="position", integrator="euler"), {}),
         ("HoverEnv", dict(action_type="velocity", integrator="euler"), {}),
         ("NavigationEnv", dict(action_type="bodyrate", integrator="rk4", drag_random=0.1), {"random_kwargs": NAV_RK}),
         ("RacingEnv", dict(action_type="thrust", integrator="euler"), {}),
         ("HoverEnv2", dict(action_type="bodyrate", integrator="euler"), {}),
         ("NavigationEnv2", dict(action_type="bodyrate", integrator="euler"), {})]
g = th.Generator(device="cuda").manual_seed(0)
for name, dkw, kw in CASES:
    dyn = dict(dt=0.0025, ctrl_dt=0.02, ctrl_delay=True, **dkw)
    env = getattr(E, name)(num_agent_per_scene=N, seed=1, dynamics_kwargs=dyn, device="cuda:0", tensor_output=True, max_episode_steps=256, **kw)
    env.reset()
    t0 = time.time()
    ends = 0.0
    for c in range(STEPS // K):
        a = (th.rand((K, N, 4), device="cuda", generator=g) * 2 - 1).contiguous()
        obs, reward, done = env.step_n(a, fused=(c % 2 == 1))
        assert bool(th.isfinite(obs).all()) and bool(th.isfinite(reward).all()), (name, c)
        ends += float(done.float().mean())
        if c % 10 == 0:
            assert bool(th.isfinite(env._slab).all()), (name, c, "slab")
            sc = env._step_count
            assert int(sc.min()) >= 0 and int(sc.max()) <= 256, (name, c, "step_count")
    th.cuda.synchronize()
    print(f"{name:15s} {dkw}: {STEPS // K * K} steps x {N} agents ok, episode-end rate {ends / (STEPS // K):.4f}, "
          f"{N * (STEPS // K * K) / (time.time() - t0):.2e} agent-steps/s incl. action generation and checks")
    env.close()
