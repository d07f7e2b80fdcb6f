"""phase breakdown of one wave of k_ppo_rollout (cycle counters; needs a -DVF_PPO_TRACE build of the library:
python -c "from visfly_amd import _build; _build.build(force=True, extra_flags=['-DVF_PPO_TRACE'])")"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from visfly_amd import _lib
from visfly_amd.envs import NavigationEnv
from visfly_amd.ppo import PPO

N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 256
dyn = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
env = NavigationEnv(num_agent_per_scene=N, seed=1, dynamics_kwargs=dyn, device="cuda:0", max_episode_steps=256, tensor_output=True,
                    random_kwargs={"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0., 2., 1.]}}]}})
ppo = PPO(env, n_steps=T, batch_size=25600, n_epochs=1, seed=2, policy_kwargs=dict(activation_fn="relu"))
for _ in range(3):
    ev = [th.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    ppo.collect_rollouts()
    ev[1].record()
    th.cuda.synchronize()
print("collect_rollouts %.3f ms, fused=%s" % (ev[0].elapsed_time(ev[1]), ppo.fused_rollout))
L = C.CDLL(_lib.lib()._name)
out = (C.c_longlong * 8)()
L.vf_debug_ppo_trace(out)
names = ["policy chain + head shuffle", "sampler + stores + ring exchange", "control interval", "env epilogue", "buffer rows + bookkeeping"]
tot = sum(out[k] for k in range(5))
for k in range(5):
    print("%-36s %9.1f cycles/step  %5.1f %%" % (names[k], out[k] / T, 100.0 * out[k] / tot))
print("total %.1f cycles/step (s_memtime / readcyclecounter ticks: 100 MHz constant clock -> x10 ns)" % (tot / T))
