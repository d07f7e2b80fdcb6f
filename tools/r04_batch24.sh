#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04b24; mkdir -p $O; cd $GRAFT_REPO_ROOT
for w in 0 1000; do
  ROC_ACTIVE_WAIT_TIMEOUT=$w timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>&1 | grep "^{" > $O/bench_wait$w.json
  python - <<PY
import json
j=json.load(open('$O/bench_wait$w.json')); t=j['timing']
print('ROC_ACTIVE_WAIT_TIMEOUT=$w value %.4e'%j['value'], 'ms/step', round(j['ms_per_step']*1e3,3), 'at completion', round(t['ms_per_step_at_completion']*1e3,3), 'kernel', round(j['roofline']['kernel_us'],3), 'resets %.4e'%j['with_resets']['value'], 'sustained %.4e'%j['sustained']['value'])
PY
done
