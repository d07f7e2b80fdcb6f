#!/bin/bash
# SHAC: a critic update's forward + loss + reverse chain as one launch (vf_twin_q_update): tests, A/B of the iteration time
O=$GRAFT_REPO_ROOT/gpurun_out/r04b26; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_shac_gpu.py tests/test_abi.py -x -q 2>&1 | tail -8 | tee $O/pytest.txt
for f in 0 1 0 1; do
  VISFLY_AMD_FUSED_CRITIC=$f timeout 600 python bench.py --workload shac --steps 256 2>&1 | tail -1 > $O/shac_$f.json
  python -c "
import json; d=json.load(open('$O/shac_$f.json')); print('fused_critic=$f  %.4e env-steps/s  %.3f ms per iteration  frac %.3f' % (d['value'], d['s_per_iteration']*1e3, d['roofline']['frac']))" | tee -a $O/ab.txt
done
