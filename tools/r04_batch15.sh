#!/bin/bash
# r04 batch 15: k_mlp_wgrad at HBM-streaming row counts (SHAC critic 524 288 rows, BPTT horizon 1 M rows): one vs two waves per SIMD
mkdir -p gpurun_out/r04b15
for w in 1 2; do
  for wl in shac bptt; do
    VISFLY_AMD_WGRAD_WPS=$w timeout 600 python bench.py --workload $wl --steps 256 2>&1 | tail -1 > gpurun_out/r04b15/${wl}_wps$w.json
    python - <<PY
import json
d=json.load(open('gpurun_out/r04b15/${wl}_wps$w.json'))
print('$wl wps=$w', d['value'], d['s_per_iteration'])
PY
  done
done
