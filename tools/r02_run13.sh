#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r02n
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_ppo_gpu.py tests/test_ppo_golden.py tests/test_bptt_gpu.py tests/test_config_scale_gpu.py tests/test_parallel_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python bench.py --workload ppo > $O/ppo.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/ppo.log") if x.startswith("{")][-1]
d=json.loads(l)
print("ppo", d["value"], d["split_ms"], d["roofline"]["us_per_update"], d["roofline"]["frac"])
PY
