"""Reset regime of the fused env step (actions U(-1,1): ~0.8 % of the agents end an episode per step, so some wave re-spawns in
every step): launch time with the real library and with a build whose spawn_agent is free (-DVF_EXP_CHEAP_SPAWN) = the bound on
what moving the Philox draws + Euler conversion off the critical wave can give.
    python tools/exp_reset_bound.py build     (here)
    python tools/exp_reset_bound.py [alt]     (GPU box; alt = the crippled build)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ALT = os.path.join(ROOT, "tools", "libvf_cheapspawn.so")
if sys.argv[1:] == ["build"]:
    from visfly_amd import _build
    _build.build(force=True, extra_flags=["-DVF_EXP_CHEAP_SPAWN"], out=ALT)
    print("built", ALT)
    sys.exit(0)
if sys.argv[1:] == ["alt"] or os.environ.get("VF_ALT_LIB"):
    from visfly_amd import _build, _lib
    _build.LIB = _lib.LIB = os.environ.get("VF_ALT_LIB") or ALT
import torch
from visfly_amd.envs import HoverEnv
N = 65536
kw = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
env = HoverEnv(num_agent_per_scene=N, seed=42, dynamics_kwargs=kw, device="cuda:0", max_episode_steps=256, tensor_output=True, out_buffers=4)
env.reset()
g = torch.Generator(device="cuda:0").manual_seed(0)
hover = torch.tensor([-1 / 3, 0, 0, 0], device="cuda:0")
calm = (hover + (torch.rand((16, N, 4), device="cuda:0", generator=g) * 2 - 1) * 0.02).clamp(-1, 1).contiguous()
wild = (torch.rand((16, N, 4), device="cuda:0", generator=g) * 2 - 1).contiguous()
for _ in range(150):
    env.step_n(calm)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(seq, reps=60):
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(reps):
            env.step_n(seq)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (reps * seq.shape[0]))
    return best
print(f"{os.path.basename(os.environ.get('VF_ALT_LIB', '')) or ('alt' if sys.argv[1:] == ['alt'] else 'real')}: no-reset regime {timed(calm):.2f} us/step", end="; ")
for _ in range(40):
    env.step_n(wild)
dn = env._rollouts[16]["done"]
print(f"reset regime {timed(wild):.2f} us/step (episode end rate {float(dn.float().mean()):.4f}, steps with a reset {float(dn.any(dim=1).float().mean()):.2f})")
