"""env-step launch time: default build vs -fno-slp-vectorize (packed-fp32 ops off), plain vs two-wave split kernels"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
code = r'''
import sys
sys.path.insert(0, %r)
import torch
from visfly_amd import _lib
_lib.LIB = sys.argv[1]
from visfly_amd.envs import HoverEnv
kw = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
out = {}
for N in (64, 32768, 65536, 131072, 262144, 1048576):
    env = HoverEnv(num_agent_per_scene=N, dynamics_kwargs=kw, device="cuda:0", tensor_output=True, max_episode_steps=256)
    env.reset()
    a = (torch.rand((N, 4), device="cuda") * 2 - 1) * 0.02 + torch.tensor([-1 / 3, 0, 0, 0], device="cuda")
    env.time_steps(a, 50)
    out[N] = round(min(env.time_steps(a, 200) for _ in range(5)), 2)
print(out)
''' % root
if __name__ == "__main__":
    from visfly_amd import _build
    libs = {"default": _build.LIB, "noslp": os.path.join(root, "visfly_amd", "csrc", "libvf_noslp.so")}
    _build.build(force=True, extra_flags=["-fno-slp-vectorize"], out=libs["noslp"])
    for name, lib in libs.items():
        for split in ("0", "1"):
            env = dict(os.environ, VISFLY_AMD_SPLIT=split)
            r = subprocess.run([sys.executable, "-c", code, lib], capture_output=True, text=True, env=env)
            print(f"{name:8s} split={split}", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-600:])
