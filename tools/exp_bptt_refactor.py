"""BPTT with the reference's actor (td_policies.Actor, policy="MultiInputPolicy"): one update as two persistent launches
(vf_bptt_rollout / vf_bptt_reverse, actor class (b)) vs the launch-by-launch sweep; device time per update (HIP events)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd.bptt import BPTT
import visfly_amd.envs as E

DYN = dict(action_type="bodyrate", ori_output_type="quaternion", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True, comm_delay=0.06,
           action_space=(-1, 1), integrator="euler", drag_random=0)


def run(cls, N, H, fused, policy):
    env = cls(num_agent_per_scene=N, seed=1, dynamics_kwargs=dict(DYN), device="cuda:0", max_episode_steps=256, requires_grad=True,
              tensor_output=True)
    algo = BPTT(env, policy=policy, horizon=H, learning_rate=1e-3, seed=3)
    algo.fused_rollout = algo.fused_reverse = fused
    for _ in range(3):
        algo._update()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for e0, e1 in ev:
        e0.record()
        algo._update()
        e1.record()
    torch.cuda.synchronize()
    t = sorted(e0.elapsed_time(e1) for e0, e1 in ev)[len(ev) // 2]
    env.close()
    return t


for cls, N, H in ((E.HoverEnv, 16384, 64), (E.NavigationEnv, 16384, 64), (E.HoverEnv, 4096, 32)):
    print(f"{cls.__name__} {N} agents, H = {H}:")
    for policy, label in ((None, "MlpPolicy actor (state-independent log_std)"), ("MultiInputPolicy", "td_policies.Actor (two heads)")):
        a, b = run(cls, N, H, True, policy), run(cls, N, H, False, policy)
        print(f"  {label:48s} persistent {a:7.3f} ms  launch by launch {b:7.3f} ms  -> {N * H / a * 1e3:.3e} / {N * H / b * 1e3:.3e} env-steps/s")
