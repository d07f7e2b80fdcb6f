#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r02r
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_ppo_gpu.py tests/test_ppo_golden.py tests/test_bptt_gpu.py tests/test_checkpoint.py -m gpu -q -x 2>&1 | tail -5
for c in 0 1; do
VISFLY_AMD_MLP_CHAIN16=$c timeout 300 python bench.py --workload bptt > $O/bptt_$c.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/bptt_$c.log") if x.startswith("{")][-1]
d=json.loads(l)
print("chain16=$c bptt", "%.4e"%d["value"], d.get("split_ms"))
PY
done
