#!/bin/bash
# r04 GPU batch 4: sub-step tape (k_bptt_rollout -> k_bptt_reverse), BPTT.learn loop fixture with the reference's actor
O=$GRAFT_REPO_ROOT/gpurun_out/r04b4; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_bptt_gpu.py tests/test_shac_gpu.py tests/test_abi.py tests/test_config_scale_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
for m in 0 1; do echo "== VISFLY_AMD_SUBSTEP_TAPE=$m" >> $O/bptt_phases.txt; VISFLY_AMD_SUBSTEP_TAPE=$m timeout 400 python tools/exp_bptt_phases.py 2>&1 | grep -v amdgpu >> $O/bptt_phases.txt; done
cat $O/bptt_phases.txt
for m in 0 1; do VISFLY_AMD_SUBSTEP_TAPE=$m timeout 300 python bench.py --workload bptt --steps 128 2>&1 | grep -v amdgpu > $O/bench_bptt_tape$m.txt; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04b4/bench_*.txt')):
    for l in open(f):
        if l.startswith('{'):
            j=json.loads(l); print(os.path.basename(f), j['value'], j['roofline'].get('frac'))
PY
