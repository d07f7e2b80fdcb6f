"""learning-curve soak of the PPO loop on NavigationEnv (close target so that successes are reachable quickly)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd.envs import NavigationEnv
from visfly_amd.ppo import PPO
N = 8192
spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0.5, 1., 0.5]}}]}}
env = NavigationEnv(num_agent_per_scene=N, seed=1, dynamics_kwargs=dict(action_type="bodyrate", integrator="euler", dt=0.0025,
                    ctrl_dt=0.02, ctrl_delay=True), random_kwargs=spawn, device="cuda:0", max_episode_steps=128, target=[4., 0., 1.5])
# third argument "pi=128,128:vf=32": a net_arch without a built-in chain class (compiled on first use, visfly_amd/_jit.py)
kw = dict(policy_kwargs=dict(activation_fn="relu"))
if len(sys.argv) > 2 and sys.argv[2] == "default":       # PPO(env) exactly as the reference policy class defaults it: Tanh trunks (policies.py:108)
    kw = {}
elif len(sys.argv) > 2:
    arch = {k: [int(x) for x in v.split(",")] for k, v in (p.split("=") for p in sys.argv[2].split(":"))}
    kw = dict(policy_kwargs=dict(features_extractor_class="StateTargetExtractor", activation_fn="ReLU", net_arch=arch,
                                 features_extractor_kwargs=dict(net_arch=dict(state=dict(layer=[128, 64]), target=dict(layer=[128, 64])))))
ppo = PPO(env, n_steps=128, batch_size=25600, n_epochs=5, learning_rate=3e-4, seed=0, **kw)
print("policy:", ppo.policy.spec, "generated chain class" if ppo.policy.chain_jit else "built-in chain class")
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    ppo.learn(128 * N)
    l = ppo.logs
    print(f"it {it:3d} ep_rew {l.get('rollout/ep_rew_mean', float('nan')):8.3f} len {l.get('rollout/ep_len_mean', float('nan')):6.1f} "
          f"success {l.get('rollout/ep_success_rate', float('nan')):.3f} v_loss {l['train/value_loss']:.4f} kl {l['train/approx_kl']:.4f} "
          f"fps {l['time/fps']:.2e}")
