#!/bin/bash
# state of HEAD after the four-lanes-per-agent BPTT launches: full GPU suite, the driver's bench command, BPTT phases, trainer legs
O=$GRAFT_REPO_ROOT/gpurun_out/r04b17; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; tail -3 $O/pytest_all.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu > $O/bench_default.txt
timeout 300 python tools/exp_bptt_phases.py > $O/bptt_phases.txt 2>&1; cat $O/bptt_phases.txt
for wl in bptt shac ppo; do
  steps=256; [ $wl = bptt ] && steps=128
  timeout 600 python bench.py --workload $wl --steps $steps 2>&1 | tail -1 > $O/$wl.json
done
python - <<'PY'
import json,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04b17/'
for l in open(O+'bench_default.txt'):
    if l.startswith('{'):
        j=json.loads(l); print('default value %.4e'%j['value'], 'ms', j['ms_per_step'], 'frac', round(j['roofline']['frac'],3), 'kernel_us', j['roofline'].get('kernel_us'))
        for k,v in j.get('secondary',{}).items(): print('   ', k, '%.4e'%v['value'], round(v['roofline']['frac'],3))
for wl in ('bptt','shac','ppo'):
    try:
        d=json.load(open(O+wl+'.json')); print(wl, '%.4e'%d['value'], d.get('s_per_iteration'), round(d['roofline']['frac'],3))
    except Exception as e: print(wl, 'failed', e)
PY
