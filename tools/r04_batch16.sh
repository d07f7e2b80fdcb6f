#!/bin/bash
mkdir -p gpurun_out/r04b16
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r04b16/pytest.txt
for wl in shac bptt ppo; do
  timeout 600 python bench.py --workload $wl --steps 256 2>&1 | tail -1 > gpurun_out/r04b16/$wl.json
  python - <<PY
import json
d=json.load(open('gpurun_out/r04b16/$wl.json'))
print('$wl', d['value'], d['s_per_iteration'], d['roofline']['frac'])
PY
done
