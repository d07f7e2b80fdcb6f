"""device time of the phases of one BPTT update at the configs[4] shard: forward half, reverse half, weight gradients, apply --
persistent launches (vf_bptt_rollout / vf_bptt_reverse) vs launch by launch"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if len(sys.argv) > 2:            # another build of the library (e.g. -DVF_PPO_TRACE: python -c "from visfly_amd import _build; _build.build(force=True, extra_flags=['-DVF_PPO_TRACE'], out='/path/x.so')")
    from visfly_amd import _build, _lib
    _build.LIB = _lib.LIB = sys.argv[2]
from visfly_amd.bptt import BPTT
from visfly_amd.envs import RacingEnv
N, H = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 64
dkw = dict(action_type="thrust", integrator=os.environ.get("VF_EXP_INTEG", "euler"), dt=float(os.environ.get("VF_EXP_DT", "0.0025")), ctrl_dt=0.02, ctrl_delay=True)   # VF_EXP_DT=0.005 / 0.01: 4 / 2 sub-steps per interval


def timed(fn, acc, key):
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        acc.setdefault(key, []).append((e0, e1))
        return r
    return w


for fwd, rev in ((True, True), (True, False), (False, False)):
    env = RacingEnv(num_agent_per_scene=N, seed=42, dynamics_kwargs=dkw, device="cuda:0", max_episode_steps=256, requires_grad=True, tensor_output=True)
    algo = BPTT(env, horizon=H, gamma=0.99, learning_rate=1e-3, seed=0)
    algo.fused_rollout, algo.fused_reverse = fwd, rev
    for _ in range(3):
        algo._update()
    acc = {}
    env.rollout_policy = timed(env.rollout_policy, acc, "forward (persistent)")
    env.reverse_policy = timed(env.reverse_policy, acc, "reverse (persistent)")
    algo.policy.weight_grad_slots = timed(algo.policy.weight_grad_slots, acc, "weight gradients")
    algo._apply = timed(algo._apply, acc, "clip + Adam + detach")
    algo._grad_reverse_sweep = timed(algo._grad_reverse_sweep, acc, "whole sweep")
    for _ in range(10):
        algo._update()
    torch.cuda.synchronize()
    print(f"N={N} H={H} fused forward={fwd} reverse={rev}")
    for k, evs in acc.items():
        ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
        print(f"   {k:24s} {ms:7.3f} ms" + (f"  = {ms / H * 1e3:6.1f} us per step" if "persistent" in k else ""))
    env.close()

# phase split inside the persistent reverse sweep (needs a -DVF_PPO_TRACE build; silently skipped otherwise)
try:
    import ctypes as _C
    from visfly_amd import _lib as _l
    _L = _C.CDLL(_l.lib()._name)
    _o = (_C.c_longlong * 8)()
    _L.vf_debug_rev_trace(_o)
    print("k_bptt_reverse, one wave, cycles per step: adjoint of the env step %.0f, reverse chain %.0f" % (_o[0] / 64.0, _o[1] / 64.0))
except AttributeError:
    pass

