import sys, time, torch
sys.path.insert(0, "/root/repo")
from visfly_amd.envs import RacingEnv
from visfly_amd.bptt import BPTT
env = RacingEnv(num_agent_per_scene=16384, seed=1, device="cuda:0", tensor_output=True, max_episode_steps=256,
                dynamics_kwargs=dict(action_type="thrust", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True))
alg = BPTT(env, horizon=64, seed=0)
for _ in range(3):
    alg._update()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); alg._update(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms")
if len(sys.argv) > 1 and sys.argv[1] == "--profile":
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        alg._update()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
