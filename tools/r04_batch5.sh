#!/bin/bash
# r04 GPU batch 5: SAC-actor chain classes (tests, SHAC timing chain vs block-tile), full default bench line
O=$GRAFT_REPO_ROOT/gpurun_out/r04b5; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_ppo_gpu.py tests/test_shac_gpu.py tests/test_bptt_gpu.py -x -q -m gpu -k "sac or shac or loop or reference_actor or chain" > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
for m in 1 0; do VISFLY_AMD_MLP_CHAIN=$m timeout 300 python bench.py --workload shac --steps 256 2>&1 | grep -v amdgpu > $O/bench_shac_chain$m.txt; done
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu > $O/bench_default.txt
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04b5/bench_*.txt')):
    for l in open(f):
        if l.startswith('{'):
            j=json.loads(l); print(os.path.basename(f), j['value'], j.get('s_per_iteration'), j['roofline'].get('frac'))
            if 'secondary' in j:
                for k,v in j['secondary'].items(): print('   secondary', k, v and v.get('value'), v and v['roofline'].get('frac'))
PY
