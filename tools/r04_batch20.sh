#!/bin/bash
# k_env_step / k_dyn_step with preloaded kernel arguments + one-batch constant prefetch: full GPU suite, the driver's bench command, A/B, timeline
O=$GRAFT_REPO_ROOT/gpurun_out/r04b20; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; grep -E "passed|failed|error" $O/pytest_all.txt | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu > $O/bench_default.txt
python - <<'PY'
import json,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04b20/'
for l in open(O+'bench_default.txt'):
    if l.startswith('{'):
        j=json.loads(l); print('default value %.4e'%j['value'], 'ms', j['ms_per_step'], 'frac', round(j['roofline']['frac'],3), 'kernel_us', j['roofline'].get('kernel_us'), 'dyn_only', j['roofline']['dyn_only']['kernel_us'], 'sustained %.3e'%j['sustained']['value'], 'resets %.3e x%.3f'%(j['with_resets']['value'], j['with_resets']['kernel_us_vs_headline']))
        for k,v in j.get('secondary',{}).items(): print('   ', k, '%.4e'%v['value'], round(v['roofline']['frac'],3))
PY
VF_ALT_LIB=$PWD/tools/tmp/libvf_nopf.so timeout 300 python tools/exp_env_quad.py 65536 2>&1 | grep QUAD | sed 's/^/before  /' | tee -a $O/ab.txt
timeout 300 python tools/exp_env_quad.py 65536 2>&1 | grep QUAD | sed 's/^/after   /' | tee -a $O/ab.txt
VF_ALT_LIB=$PWD/tools/tmp/libvf_trace.so timeout 300 python tools/exp_env_timeline.py > $O/timeline.txt 2>&1
