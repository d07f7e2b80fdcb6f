"""is the PPO update loop host- or GPU-bound?  host time to ENQUEUE n minibatch updates vs time until the GPU is done"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd.envs import NavigationEnv
from visfly_amd.ppo import PPO
DEV = "cuda:0"
N = 32768
spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0., 2., 1.]}}]}}
env = NavigationEnv(num_agent_per_scene=N, seed=42, dynamics_kwargs=dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True),
                    random_kwargs=spawn, device=DEV, max_episode_steps=256)
ppo = PPO(env, n_steps=32, batch_size=25600, n_epochs=1, learning_rate=1e-4, seed=0)
ppo.collect_rollouts()
buf = ppo.buf
flat = {"actions": buf.actions.view(-1, 4), "old_lp": buf.log_probs.view(-1), "adv": buf.advantages.view(-1), "ret": buf.returns.view(-1)}
flat.update({"obs:" + k: buf.obs[k].view(-1, buf.obs[k].shape[-1]) for k in ppo.obs_keys})
mb = {k: v[:25600] for k, v in flat.items()}
acc = torch.zeros(16, device=DEV)
for _ in range(20): ppo._minibatch_update(mb, acc)
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n): ppo._minibatch_update(mb, acc)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e6 * (t1 - t0) / n:.1f} us/update   until GPU done {1e6 * (t2 - t0) / n:.1f} us/update")
