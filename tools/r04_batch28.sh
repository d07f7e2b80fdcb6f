#!/bin/bash
# PPO: the next epoch's shuffle (randperm + row gather) on a side stream beside the current epoch's optimiser steps: tests + A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r04b28; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_ppo_gpu.py tests/test_parallel_gpu.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
for f in 0 1 0 1; do
  VISFLY_AMD_OVERLAP_SHUFFLE=$f timeout 600 python bench.py --workload ppo --steps 256 2>&1 | tail -1 > $O/ppo_$f.json
  python -c "
import json; d=json.load(open('$O/ppo_$f.json')); print('overlap_shuffle=$f  %.4e env-steps/s  %.2f ms per iteration  frac %.3f' % (d['value'], d['s_per_iteration']*1e3, d['roofline']['frac']), d.get('split_ms'))" | tee -a $O/ab.txt
done
