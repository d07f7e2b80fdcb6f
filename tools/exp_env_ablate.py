"""r02: build and run tools/env_step_probe.hip (ablated forms of the env-step launch) -- `python tools/exp_env_ablate.py build`
here (cross-compiles, writes the cfg blob), `tools/env_step_probe tools/cfg_env_hover.bin` on the GPU box."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfly_amd import _lib  # noqa: E402
from visfly_amd.constants import derive_constants  # noqa: E402

consts = derive_constants(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
d = _lib.DynCfg.from_dict(consts)
e = _lib.EnvCfg()
e.kind, e.max_episode_steps, e.is_collision_reset, e.n_spawn = 0, 256, 1, 1
for k in range(3):
    e.bbox_lo[k], e.bbox_hi[k], e.target[k] = (-30., -30., 0.)[k], (30., 30., 8.)[k], (1., 0., 1.5)[k]
    e.spawn[0].pos_mean[k], e.spawn[0].pos_half[k] = (1., 0., 1.5)[k], (1., 1., .5)[k]
e.uav_radius, e.success_radius, e.seed = 0.1, 0.5, 42
with open(os.path.join(ROOT, "tools", "cfg_env_hover.bin"), "wb") as f:
    f.write(bytes(d))
    f.write(bytes(e))
if len(sys.argv) > 1 and sys.argv[1] == "build":
    for mode in (0, 1, 2, 3):          # VF_STORE_MODE variants: plain | sc1 | nt | sc0 sc1
        out = os.path.join(ROOT, "tools", "env_step_probe" + ("" if mode == 0 else f"_st{mode}"))
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-w",
                               f"-DVF_STORE_MODE={mode}", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "visfly_amd", "csrc"),
                               os.path.join(ROOT, "tools", "env_step_probe.hip"), "-o", out])
        print("built", out)
    out = os.path.join(ROOT, "tools", "env_step_probe_fast")          # VF_FAST_EXACT=1: exact fast paths for sqrt and x / m
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-w",
                           "-DVF_FAST_EXACT=1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "visfly_amd", "csrc"),
                           os.path.join(ROOT, "tools", "env_step_probe.hip"), "-o", out])
    print("built", out)
