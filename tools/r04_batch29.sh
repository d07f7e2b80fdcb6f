#!/bin/bash
# SHAC: the horizon's 32 target-critic passes + 32 accumulate launches as one launch each (vf_mlp_forward_steps, vf_shac_accumulate_horizon): tests, A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r04b29; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_shac_gpu.py -x -q 2>&1 | tail -4 | tee $O/pytest.txt
for f in 0 1 0 1; do
  VISFLY_AMD_BATCH_TARGETS=$f timeout 600 python bench.py --workload shac --steps 256 2>&1 | tail -1 > $O/shac_$f.json
  python -c "
import json; d=json.load(open('$O/shac_$f.json')); print('batch_targets=$f  %.4e env-steps/s  %.3f ms per iteration  frac %.3f' % (d['value'], d['s_per_iteration']*1e3, d['roofline']['frac']))" | tee -a $O/ab.txt
done
