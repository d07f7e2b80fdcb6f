import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
from visfly_amd.envs import HoverEnv
N = 65536
env = HoverEnv(num_agent_per_scene=N, seed=1, device="cuda:0", tensor_output=False, max_episode_steps=256,
               dynamics_kwargs=dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True))
env.reset()
a = np.tile(np.array([-1 / 3, 0, 0, 0], np.float32), (N, 1))
for _ in range(20):
    env.step(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 200
for _ in range(K):
    obs, r, d, info = env.step(a)          # numpy action in, numpy reward / done out (reference convention)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"tensor_output=False, host action (N,4) in, numpy reward/done + host obs out: {N * K / el:.3e} agent-steps/s ({1e6 * el / K:.1f} us/step), obs type {type(obs['state'])}")
