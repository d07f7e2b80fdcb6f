"""functional check of the first-order trainers on the final kernels: BPTT (the reference's actor) and SHAC on HoverEnv, with the default
network shapes (built-in chain classes), a non-default net_arch (generated actor / twin-critic classes, horizons from the BPTT plugin) and
SHAC with share_features_extractor=True -- mean reward per step of the horizon and the losses over the iterations
    python tools/exp_td_learn.py [iterations]"""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd.bptt import BPTT
from visfly_amd.envs import HoverEnv
from visfly_amd.shac import SHAC

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
N, H = 4096, 32
DYN = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [1.0, 1.0, 0.5]}}]}}


def pk(widths, share=False):
    return dict(features_extractor_class="StateExtractor", features_extractor_kwargs={"net_arch": {"state": {"layer": [128, 64]}}},
                net_arch=dict(pi=widths, qf=widths), activation_fn="relu", share_features_extractor=share)


for name, cls, kw in (("BPTT default", BPTT, dict(policy="MultiInputPolicy")),
                      ("BPTT pi=[128,128] (generated class)", BPTT, dict(policy="MultiInputPolicy", policy_kwargs=pk([128, 128]))),
                      ("SHAC default", SHAC, dict()),
                      ("SHAC pi=qf=[128,128] (generated classes)", SHAC, dict(policy_kwargs=pk([128, 128]))),
                      ("SHAC default, share_features_extractor=True", SHAC, dict(policy_kwargs=pk([64, 64], True)))):
    env = HoverEnv(num_agent_per_scene=N, seed=1, dynamics_kwargs=dict(DYN), random_kwargs=spawn, device="cuda:0", max_episode_steps=128,
                   requires_grad=True, tensor_output=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        algo = cls(env, horizon=H, learning_rate=1e-3, seed=0, **kw)
        rows = []
        for it in range(iters):
            algo.learn(H * N)
            if it % max(1, iters // 6) == 0 or it == iters - 1:
                logs = algo.flush_logs() if hasattr(algo, "flush_logs") else algo.logs
                rew = algo._buf["reward"].mean().item() if getattr(algo, "_buf", None) else algo._last_rollout.get("reward", torch.zeros(1)).float().mean().item()
                rows.append(f"it {it:3d}: actor_loss {logs['train/actor_loss']:9.4f}" + (f"  critic_loss {logs['train/critic_loss']:8.5f}  mean reward/step {rew:7.4f}" if "train/critic_loss" in logs else ""))
    fall = [str(x.message)[:80] for x in w if "falling back" in str(x.message)]
    print(f"== {name}: chain_jit={algo.policy.chain_jit}, fallback warnings {len(fall)}, {algo.logs.get('time/fps', 0):.3g} env-steps/s")
    for r in rows:
        print("   ", r)
    env.close()
