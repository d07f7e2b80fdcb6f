#!/bin/bash
# r04 GPU batch 9: chain-kernel stores with per-layer pinned bases (no per-store scalar loads / exec branches): tests + timing of all three trainers
O=$GRAFT_REPO_ROOT/gpurun_out/r04b9; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 300 python bench.py --workload ppo --steps 256 2>&1 | grep -v amdgpu > $O/bench_ppo.txt
timeout 300 python bench.py --workload bptt --steps 128 2>&1 | grep -v amdgpu > $O/bench_bptt.txt
timeout 300 python bench.py --workload shac --steps 256 2>&1 | grep -v amdgpu > $O/bench_shac.txt
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04b9/bench_*.txt')):
    for l in open(f):
        if l.startswith('{'):
            j=json.loads(l); print(os.path.basename(f), j['value'], j.get('s_per_iteration'), j['roofline'].get('frac'), j['roofline'].get('us_per_update'))
PY
timeout 400 python tools/exp_bptt_phases.py 2>&1 | grep -v amdgpu | head -6
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/pm3 -- python $R/tools/exp_ppo_update_one.py 25600 200 > $O/log3.txt 2>&1
python $R/tools/prof_summary.py $(ls /tmp/pm3/*/*kernel_stats.csv | head -1) $O/update_one_kernel_stats.txt "python tools/exp_ppo_update_one.py 25600 200" > /dev/null 2>&1
head -7 $O/update_one_kernel_stats.txt
