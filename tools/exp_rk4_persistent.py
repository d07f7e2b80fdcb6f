"""BASELINE configs[2]'s dynamics (RK4 + ctrl_delay + per-agent drag randomisation) under the trainers' closed loops: the persistent
launches (r04: INTEG = RK4 instances of k_ppo_rollout / k_bptt_rollout / k_bptt_reverse) against the launch-by-launch fallback they
replace.  HIP events around collect_rollouts / one BPTT update, median of 5."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd.bptt import BPTT
from visfly_amd.envs import HoverEnv, NavigationEnv
from visfly_amd.ppo import PPO

DYN = dict(action_type="bodyrate", integrator="rk4", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True, drag_random=0.5)
SPAWN = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0., 2., 1.]}}]}}


def med(fn, n=5):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[n // 2]


print("NavigationEnv 32 768 agents, RK4 + drag randomisation, PPO collect_rollouts (256 steps):")
for fused in (True, False):
    env = NavigationEnv(num_agent_per_scene=32768, seed=42, dynamics_kwargs=dict(DYN), random_kwargs=SPAWN, device="cuda:0", max_episode_steps=256)
    ppo = PPO(env, n_steps=256, batch_size=25600, n_epochs=1, seed=0, policy_kwargs=dict(activation_fn="relu"))
    ppo.fused_rollout = fused
    ppo.collect_rollouts()
    ms = med(ppo.collect_rollouts)
    assert ppo.fused_rollout is fused
    print(f"  {'one persistent launch' if fused else 'launch by launch     '}  {ms:8.2f} ms  = {ms / 256 * 1e3:6.1f} us per step  {32768 * 256 / ms * 1e3:.3e} env-steps/s")
    env.close()
print("HoverEnv 16 384 agents, RK4 + drag randomisation, BPTT update (H = 64):")
for fused in (True, False):
    env = HoverEnv(num_agent_per_scene=16384, seed=42, dynamics_kwargs=dict(DYN), device="cuda:0", max_episode_steps=256, requires_grad=True,
                   tensor_output=True)
    algo = BPTT(env, horizon=64, gamma=0.99, learning_rate=1e-3, seed=0)
    algo.fused_rollout = algo.fused_reverse = fused
    algo.learn(64 * 16384)
    ms = med(lambda: algo.learn(64 * 16384))
    print(f"  {'persistent forward + reverse' if fused else 'launch by launch            '}  {ms:8.2f} ms per update  {16384 * 64 / ms * 1e3:.3e} env-steps/s")
    env.close()
