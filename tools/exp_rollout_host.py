"""host vs GPU time of PPO.collect_rollouts (NavigationEnv, 32768 agents, 256 steps) + cProfile of the host path"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd.envs import NavigationEnv
from visfly_amd.ppo import PPO
spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1.0, 0.0, 1.5], "half": [0.0, 2.0, 1.0]}}]}}
env = NavigationEnv(num_agent_per_scene=32768, seed=1, device="cuda:0", tensor_output=True, max_episode_steps=256, random_kwargs=spawn,
                    dynamics_kwargs=dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True))
ppo = PPO(env, n_steps=256, batch_size=25600, n_epochs=5, policy_kwargs=dict(activation_fn="relu"))
ppo.collect_rollouts()
torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter(); ppo.collect_rollouts(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rollout: host enqueue {1e3 * (t1 - t0):.1f} ms, total {1e3 * (t2 - t0):.1f} ms")
if len(sys.argv) > 1 and sys.argv[1] == "--profile":
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    ppo.collect_rollouts()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
