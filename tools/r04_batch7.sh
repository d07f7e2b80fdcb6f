#!/bin/bash
# r04 GPU batch 7: completion-wait mode of the timed region (HSA_ENABLE_INTERRUPT) on the headline's --steps 20 regions; SAC tests
O=$GRAFT_REPO_ROOT/gpurun_out/r04b7; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for m in 1 0; do HSA_ENABLE_INTERRUPT=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --sustain-s 1 2>&1 | grep -v amdgpu > $O/bench_int$m.$rep.txt; done; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04b7/bench_int*.txt')):
    for l in open(f):
        if l.startswith('{'):
            j=json.loads(l); t=j['timing']; print(os.path.basename(f), 'value %.4e'%j['value'], 'ms/step', round(j['ms_per_step']*1e3,2), 'at completion', round(t['ms_per_step_at_completion']*1e3,2), 'event', round(t['event_us_per_step'],2), 'host', round(t['host_us_per_step'],2), 'resets %.4e'%j['with_resets']['value'])
PY
timeout 900 python -m pytest tests/test_ppo_gpu.py -x -q -m gpu -k "sac" 2>&1 | tail -3
