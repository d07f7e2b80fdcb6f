#!/bin/bash
# r04 GPU batch 2: k_mlp_wgrad with slab-bounded buffer loads (A/B one vs two waves per SIMD) + the whole GPU suite on the 256-step fixtures
O=$GRAFT_REPO_ROOT/gpurun_out/r04b2; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
for w in 1 2; do VISFLY_AMD_WGRAD_WPS=$w timeout 200 python tools/exp_wgrad_mall.py 2>&1 | grep -v amdgpu > $O/wgrad_mall_wps$w.txt; done
for w in 1 2; do VISFLY_AMD_WGRAD_WPS=$w timeout 300 python bench.py --workload ppo --steps 256 2>&1 | grep -v amdgpu > $O/bench_ppo_wps$w.txt; done
timeout 300 python bench.py --workload bptt --steps 128 2>&1 | grep -v amdgpu > $O/bench_bptt.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
tail -5 $O/pytest_all.txt; cat $O/wgrad_mall_wps*.txt
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04b2/bench_*.txt')):
    for l in open(f):
        if l.startswith('{'):
            j=json.loads(l); print(os.path.basename(f), j['value'], j.get('split_ms'), j['roofline'].get('us_per_update'), j['roofline'].get('frac'))
PY
