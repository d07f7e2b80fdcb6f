#!/bin/bash
# k_mlp_wgrad in the streaming regime: waves per layer by bytes per row instead of by MFMA tiles -- SHAC / BPTT A/B + tests
O=$GRAFT_REPO_ROOT/gpurun_out/r04b30; mkdir -p $O; cd $GRAFT_REPO_ROOT
for b in tiles bytes tiles bytes; do
  for w in shac bptt; do
  steps=256; [ $w = bptt ] && steps=128
  VISFLY_AMD_WGRAD_BALANCE=$b timeout 600 python bench.py --workload $w --steps $steps 2>&1 | tail -1 > $O/${w}_$b.json
  python -c "
import json; d=json.load(open('$O/${w}_$b.json')); print('$w balance=$b  %.4e env-steps/s  %.3f ms per iteration  frac %.3f' % (d['value'], d['s_per_iteration']*1e3, d['roofline']['frac']))" | tee -a $O/ab.txt
  done
done
timeout 900 python -m pytest tests/test_shac_gpu.py tests/test_bptt_gpu.py tests/test_ppo_gpu.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
