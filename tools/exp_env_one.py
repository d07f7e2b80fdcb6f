"""run the fused HoverEnv step kernel a few times at a given N (for rocprofv3 --pmc passes); `fused` as third argument runs the
same number of steps as ONE vf_env_rollout_fused launch per 8 steps instead"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get('VF_ALT_LIB'):
    from visfly_amd import _build, _lib
    _build.LIB = _lib.LIB = os.environ['VF_ALT_LIB']
from visfly_amd.envs import HoverEnv
N = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fused = len(sys.argv) > 3 and sys.argv[3] == "fused"
kw = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
env = HoverEnv(num_agent_per_scene=N, dynamics_kwargs=kw, device="cuda:0", tensor_output=True, max_episode_steps=256)
env.reset()
a = (torch.rand((N, 4), device="cuda") * 2 - 1) * 0.02 + torch.tensor([-1 / 3, 0, 0, 0], device="cuda")
if fused:
    A = a.unsqueeze(0).repeat(8, 1, 1).contiguous()
    for _ in range(iters):
        env.step_n(A, fused=True)
else:
    for _ in range(iters):
        env.step(a)
torch.cuda.synchronize()
