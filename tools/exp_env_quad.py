"""HoverEnv.step launch time, one lane per agent vs four lanes per agent (VISFLY_AMD_ENV_QUAD=1 on a -DVF_EXP_ENV_QUAD build given by
VF_ALT_LIB), headline and reset regimes, vs agents and vs sub-steps per interval (VF_EXP_DT)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get('VF_ALT_LIB'):     # the measurement build: python -c "from visfly_amd import _build; _build.build(force=True, extra_flags=['-DVF_EXP_ENV_QUAD'], out='/tmp/libvf_quad.so')"
    from visfly_amd import _build, _lib
    _build.LIB = _lib.LIB = os.environ['VF_ALT_LIB']
from visfly_amd.envs import HoverEnv

kw = dict(action_type="bodyrate", integrator="euler", dt=float(os.environ.get("VF_EXP_DT", "0.0025")), ctrl_dt=0.02, ctrl_delay=True)   # VF_EXP_DT=0.005 / 0.01: 4 / 2 sub-steps
Ns = [int(x) for x in sys.argv[1:]] or [16384, 65536, 262144, 1048576]
for N in Ns:
    env = HoverEnv(num_agent_per_scene=N, dynamics_kwargs=kw, device="cuda:0", tensor_output=True, max_episode_steps=256)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    a = (torch.rand((N, 4), device="cuda", generator=g) * 2 - 1) * 0.02 + torch.tensor([-1 / 3, 0, 0, 0], device="cuda")
    env.time_steps(a, 60)
    us = min(env.time_steps(a, 200) for _ in range(5))
    ar = torch.rand((N, 4), device="cuda", generator=g) * 2 - 1
    env.time_steps(ar, 300)
    usr = min(env.time_steps(ar, 200) for _ in range(5))
    print(f"dt={kw['dt']} QUAD={os.environ.get('VISFLY_AMD_ENV_QUAD', '0')} N={N:8d}  headline {us:8.2f} us/launch  reset regime {usr:8.2f} us  {N / us * 1e6:.3e} agent-steps/s", flush=True)
    env.close()
