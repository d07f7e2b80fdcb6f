"""can the horizon's weight-gradient launch hide inside the persistent reverse sweep?  Timing only: the reverse sweep of a BPTT update and the
weight-gradient launch of (another copy of) the same rows, alone and concurrently on two streams (the results of the concurrent run are not used)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd.bptt import BPTT
from visfly_amd.envs import RacingEnv

N, H = 16384, 64
dkw = dict(action_type="thrust", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
env = RacingEnv(num_agent_per_scene=N, seed=42, dynamics_kwargs=dkw, device="cuda:0", max_episode_steps=256, requires_grad=True, tensor_output=True)
algo = BPTT(env, horizon=H, gamma=0.99, learning_rate=1e-3, seed=0)
for _ in range(3):
    algo._update()
pol = algo.policy
calls = {}
orig_rev, orig_wg = env.reverse_policy, pol.weight_grad_slots


def grab(name, fn):
    def w(*a, **k):
        calls[name] = (a, k)
        return fn(*a, **k)
    return w


env.reverse_policy = grab("rev", orig_rev)
pol.weight_grad_slots = grab("wg", orig_wg)
algo._update()
env.reverse_policy, pol.weight_grad_slots = orig_rev, orig_wg
torch.cuda.synchronize()
ra, rk = calls["rev"]
wa, wk = calls["wg"]
# the reverse sweep needs its tape position: re-run it on the recorded horizon (env._tape_t was reset by the update's detach)
env._tape_t = H
env._substep_range = (0, H)
s2 = torch.cuda.Stream()


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def rev():
    env._tape_t = H
    assert orig_rev(*ra, **rk)


def wg():
    orig_wg(*wa, **wk)


def both():
    ev = torch.cuda.Event()
    ev.record()
    rev()
    with torch.cuda.stream(s2):
        s2.wait_event(ev)
        wg()
    torch.cuda.current_stream().wait_stream(s2)


def half_both():      # what the split would look like in time: reverse, then [reverse || weight gradients]
    rev()
    both()


t_rev, t_wg, t_both = timed(rev), timed(wg), timed(both)
print(f"reverse sweep alone {t_rev:.3f} ms   weight gradients alone {t_wg:.3f} ms   sum {t_rev + t_wg:.3f}   concurrent on two streams {t_both:.3f} ms")
