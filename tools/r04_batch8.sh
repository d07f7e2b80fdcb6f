#!/bin/bash
# r04 GPU batch 8: twin-critic chain class (tests + SHAC timing), adjoint division trim (BPTT tests + phases), full GPU suite
O=$GRAFT_REPO_ROOT/gpurun_out/r04b8; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_ppo_gpu.py -x -q -m gpu -k "critic or sac" > $O/pytest_critic.txt 2>&1; tail -15 $O/pytest_critic.txt
timeout 900 python -m pytest tests/test_shac_gpu.py tests/test_bptt_gpu.py -x -q -m gpu > $O/pytest_shac_bptt.txt 2>&1; tail -5 $O/pytest_shac_bptt.txt
for m in 1 0; do VISFLY_AMD_MLP_CHAIN=$m timeout 300 python bench.py --workload shac --steps 256 2>&1 | grep -v amdgpu > $O/bench_shac_chain$m.txt; done
timeout 300 python bench.py --workload bptt --steps 128 2>&1 | grep -v amdgpu > $O/bench_bptt.txt
timeout 400 python tools/exp_bptt_phases.py 2>&1 | grep -v amdgpu | head -8 > $O/bptt_phases.txt; cat $O/bptt_phases.txt
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04b8/bench_*.txt')):
    for l in open(f):
        if l.startswith('{'):
            j=json.loads(l); print(os.path.basename(f), j['value'], j.get('s_per_iteration'), j['roofline'].get('frac'))
PY
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; tail -5 $O/pytest_all.txt
