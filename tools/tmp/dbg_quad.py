import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from visfly_amd.bptt import BPTT
import visfly_amd.envs as E
DYN = dict(action_type="bodyrate", ori_output_type="quaternion", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True, comm_delay=0.06,
           action_space=(-1, 1), integrator=os.environ.get("INTEG", "euler"), drag_random=0)
N, H = 16, int(os.environ.get("H", "1"))
res = []
for fused in (True, False):
    env = E.HoverEnv(num_agent_per_scene=N, seed=5, dynamics_kwargs=dict(DYN), device="cuda:0", max_episode_steps=100, requires_grad=True, tensor_output=True)
    algo = BPTT(env, horizon=H, learning_rate=1e-3, seed=9)
    algo.fused_rollout = True
    algo.fused_reverse = fused
    # a few steps first so that velocities / rates are non-trivial
    for _ in range(3):
        algo._grad_reverse_sweep(); env.clear_tape()
    env._adj.normal_()     # incoming adjoint: random, same in both runs
    torch.manual_seed(0)
    env._adj.copy_(torch.randn(env._adj.shape, generator=torch.Generator().manual_seed(1)).to("cuda:0"))
    loss = algo._grad_reverse_sweep()
    G = env._adj.shape
    live = lambda x: x.transpose(-3, -4).reshape(*x.shape[:-4], x.shape[-3], -1, 4)[..., :N, :].clone()
    res.append((live(env._adj), algo.policy.grad.clone()))
    env.close()
a, b = res[0][0], res[1][0]
print("adj shape", a.shape)
for gidx in range(a.shape[0]):
    d = (a[gidx] - b[gidx]).abs().max().item()
    print("granule", gidx, "max diff", d, "fused", a[gidx][0].tolist(), "loop", b[gidx][0].tolist())
