"""r06: what the generated twin-critic class and the BPTT plugins buy on a NON-default net_arch.
    python tools/exp_generated_horizons.py <mode> [pi widths, default 128,128]
modes (one process each: a plugin that was loaded stays loaded)
    all        r06: actor horizons = the class's persistent launches (BPTT plugin), critic = generated class with the fused step
    loop       actor launch by launch on the generated chains (r05), critic = generated class
    r05        actor launch by launch on the generated chains, critic on the block-tile kernels (no critic plugin: the r05 tree's state)
    blocktile  no plugins at all
Prints BPTT(policy="MultiInputPolicy") update ms and SHAC iteration ms at HoverEnv, 16 384 agents."""
import sys
import time
import warnings

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
mode = sys.argv[1]
widths = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "128,128").split(",")]
from visfly_amd import _jit, _lib  # noqa: E402

if mode == "r05":
    orig = _jit.ensure
    _jit.ensure = lambda sh: False if (sh is not None and _jit._heads(sh) == (1, 1)) else orig(sh)
if mode == "blocktile":
    _lib.lib().vf_chain_plugin_set_enabled(0)
    _jit.ensure = lambda sh: False
    _jit.ensure_bptt = lambda sh, cfg: False
from visfly_amd.bptt import BPTT  # noqa: E402
from visfly_amd.envs import HoverEnv  # noqa: E402
from visfly_amd.shac import SHAC  # noqa: E402

DEV = "cuda:0"
DYN = dict(action_type="bodyrate", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True, integrator="euler")
N = 16384
pk = dict(features_extractor_class="StateExtractor", features_extractor_kwargs={"net_arch": {"state": {"layer": [128, 64]}}},
          net_arch=dict(pi=widths, qf=widths), activation_fn="relu", share_features_extractor=False)


def timed(algo, steps_per_iter, iters):
    algo.learn(steps_per_iter * 2)
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        t0 = time.perf_counter()
        algo.learn(steps_per_iter * iters)
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / iters)
    return sorted(best)[1]


with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    env = HoverEnv(num_agent_per_scene=N, seed=1, dynamics_kwargs=dict(DYN), device=DEV, max_episode_steps=256, requires_grad=True, tensor_output=True)
    algo = BPTT(env, policy="MultiInputPolicy", policy_kwargs=dict(pk), horizon=64, learning_rate=1e-3, seed=0)
    if mode in ("loop", "r05", "blocktile"):
        algo.fused_rollout = algo.fused_reverse = False
    t_bptt = timed(algo, 64 * N, 6)
    wa = algo.policy.n_params
    env.close()
    env = HoverEnv(num_agent_per_scene=N, seed=1, dynamics_kwargs=dict(DYN), device=DEV, max_episode_steps=256, requires_grad=True, tensor_output=True)
    algo = SHAC(env, policy="MultiInputPolicy", policy_kwargs=dict(pk), horizon=32, learning_rate=1e-3, gradient_steps=5, seed=0)
    if mode in ("loop", "r05", "blocktile"):
        algo.fused_rollout = algo.fused_reverse = False
    t_shac = timed(algo, 32 * N, 6)
    wc = algo.critic.n_params
    env.close()
fall = sorted({str(x.message)[:90] for x in w if "falling back" in str(x.message) or "no chain kernels" in str(x.message)})
f_b = 6.0 * wa * 64 * N / t_bptt / 1e12 / 157.3
f_s = (8.0 * wa + 2.0 * wc + 30.0 * wc) * 32 * N / t_shac / 1e12 / 157.3
print(f"{mode:10s} pi=qf={widths}: BPTT update {t_bptt * 1e3:7.2f} ms = {64 * N / t_bptt:.3e} env-steps/s ({f_b:.3f} of fp32-MFMA peak)   "
      f"SHAC iteration {t_shac * 1e3:7.2f} ms = {32 * N / t_shac:.3e} env-steps/s ({f_s:.3f})   plugin launches {_lib.lib().vf_chain_plugin_launches()}"
      f"   fallback warnings: {len(fall)}")
for m in fall:
    print("      ", m)
