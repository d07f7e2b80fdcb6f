"""the optimiser step of PPO on a network shape the library holds no chain instance of, with the chain plugin
compiled on first use (visfly_amd/_jit.py) and with the plugins switched off (vf_chain_plugin_set_enabled(0): block-tile kernels).
bench.py's PPO workload (NavigationEnv, 25 600 agents, batch 25 600) with fewer steps per rollout; fp32 MFMA peak 157.3 TFLOP/s;
flops per optimiser step = 6 x (weights + biases) x rows, as bench.py counts them.  Run on the GPU box: python tools/exp_chain_jit.py"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd import _lib
from visfly_amd.envs import NavigationEnv
from visfly_amd.ppo import PPO

lib = _lib.lib()
dev = "cuda:0"
dyn = dict(action_type="bodyrate", ori_output_type="quaternion", dt=0.0025, ctrl_dt=0.02, integrator="euler", drag_random=0.0)
ext = dict(features_extractor_class="StateTargetExtractor",
           features_extractor_kwargs=dict(net_arch=dict(state=dict(layer=[128, 64]), target=dict(layer=[128, 64]))), activation_fn="ReLU")
NETS = [("pi[64,64] vf[64,64] (built-in NetNav)", dict(pi=[64, 64], vf=[64, 64])),
        ("pi[128,128] vf[32]", dict(pi=[128, 128], vf=[32])),
        ("pi[128,128] vf[128,128]", dict(pi=[128, 128], vf=[128, 128])),
        ("pi[128,128,128] vf[128,128,128]", dict(pi=[128, 128, 128], vf=[128, 128, 128])),
        ("pi[32] vf[32]", dict(pi=[32], vf=[32]))]
print(f"{'net_arch':40} {'kernels':12} {'weights':>8} {'us/step':>9} {'TFLOP/s':>8} {'of peak':>8} {'rollout ms/64':>14}")
for name, arch in NETS:
    for on in (1, 0):
        if on == 0 and "built-in" in name:
            continue
        lib.vf_chain_plugin_set_enabled(on)
        env = NavigationEnv(num_agent_per_scene=25600, seed=1, device=dev, max_episode_steps=256, tensor_output=True, dynamics_kwargs=dict(dyn))
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            ppo = PPO(env, n_steps=64, batch_size=25600, n_epochs=3, learning_rate=1e-4, seed=0, policy_kwargs=dict(ext, net_arch=arch))
            ppo.learn(64 * 25600)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            n0 = ppo._opt_step
            ev[0].record()
            ppo.collect_rollouts()
            ev[1].record()
            ppo.train()
            ev[2].record()
            torch.cuda.synchronize()
        us = ev[1].elapsed_time(ev[2]) * 1e3 / (ppo._opt_step - n0)
        tf = 6.0 * ppo.policy.log_std_off * 25600 / (us * 1e-6) / 1e12
        kern = "built-in" if "built-in" in name else ("plugin" if on else "block-tile")
        warned = any("block-tile" in str(x.message) for x in w)
        print(f"{name:40} {kern:12} {ppo.policy.log_std_off:8d} {us:9.1f} {tf:8.1f} {tf / 157.3:8.3f} {ev[0].elapsed_time(ev[1]):14.2f}  {'(warned: block-tile)' if warned else ''}", flush=True)
        env.close()
        del ppo, env
lib.vf_chain_plugin_set_enabled(1)
