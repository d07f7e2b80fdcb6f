#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r02t
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_ppo_gpu.py tests/test_ppo_golden.py tests/test_config_scale_gpu.py tests/test_parallel_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
for ov in 0 1; do
VISFLY_AMD_PPO_OVERLAP=$ov timeout 300 python bench.py --workload ppo > $O/ppo_$ov.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/ppo_$ov.log") if x.startswith("{")][-1]
d=json.loads(l)
print("overlap=$ov ppo", "%.4e"%d["value"], d["split_ms"]["train"], d["config"]["logs"]["train/value_loss"], d["config"]["logs"]["train/approx_kl"])
PY
done
