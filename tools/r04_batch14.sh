#!/bin/bash
mkdir -p gpurun_out/r04b14
timeout 600 python bench.py --workload bptt --steps 128 2>&1 | tail -1 | tee gpurun_out/r04b14/bench_bptt.json
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r04b14/bench_default.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04b14/bench_default.json'))
print(d['value'], d['roofline']['frac'], {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d.get('secondary',{}).items()})
PY
