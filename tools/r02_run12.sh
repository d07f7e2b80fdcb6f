#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r02m
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_env_gpu.py tests/test_env_multistep_gpu.py tests/test_dynamics_gpu.py -m gpu -q -x 2>&1 | tail -2
python tools/exp_dyn_vs_env.py 2>&1 | grep -v amdgpu | tail -8
for st in 20 2000; do
  timeout 300 python bench.py --steps $st --warmup 5 --no-secondary --no-cpu-baseline > $O/b_${st}.log 2>&1
  python - <<PY
import json
l=[x for x in open("$O/b_${st}.log") if x.startswith("{")][-1]
d=json.loads(l)
print("steps", $st, "us/step", round(d["ms_per_step"]*1e3,3), "kernel", round(d["roofline"]["kernel_us"],3), "dyn", round(d["roofline"]["dyn_only"]["kernel_us"],3), "resets", round(d["with_resets"]["us_per_step"],3), round(d["with_resets"]["kernel_us"],3), "fused", round(d["rollout_fused"]["us_per_step"],3), "per_call", round(d["timing"]["per_call"]["ms_per_step"]*1e3,3))
PY
done
