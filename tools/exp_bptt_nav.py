"""first-order policy optimisation (BPTT) on NavigationEnv through the Nav reward adjoint: success rate of a deterministic roll-out"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visfly_amd.bptt import BPTT
from visfly_amd.envs import NavigationEnv
N = 4096
spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0.5, 1., 0.5]}}]}}
env = NavigationEnv(num_agent_per_scene=N, seed=1, dynamics_kwargs=dict(action_type="bodyrate", integrator="euler", dt=0.0025,
                    ctrl_dt=0.02, ctrl_delay=True), random_kwargs=spawn, device="cuda:0", max_episode_steps=128, target=[4., 0., 1.5],
                    tensor_output=True)
algo = BPTT(env, horizon=32, learning_rate=1e-3, seed=0)
def evaluate():
    with torch.no_grad():
        obs, ret, succ = env.reset(), 0.0, torch.zeros(N, dtype=torch.bool, device="cuda:0")
        for _ in range(128):
            mean, _ = algo.policy.forward({k: obs[k].contiguous() for k in algo.obs_keys}, save_activations=False)
            obs, r, d, _ = env.step(torch.tanh(mean))
            ret += float(r.mean())
            succ |= env._ep_flags.bool() & d & ((env._ep_flags & 1) != 0)
    env.detach()
    return ret, float(succ.float().mean())
print("before", evaluate())
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    algo.learn(32 * N * 25)
    print(it, "after", evaluate(), "fps %.2e" % algo.logs["time/fps"])
