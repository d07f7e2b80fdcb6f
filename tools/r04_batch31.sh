#!/bin/bash
# k_wgrad_fold with 16 instead of 4 waves per block (the partial rows split 16 ways): PPO A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r04b31; mkdir -p $O; cd $GRAFT_REPO_ROOT
for v in 4 16 4 16; do
  L=""; [ $v = 16 ] && L=$PWD/tools/tmp/libvf_fold16.so
  VF_ALT_LIB=$L timeout 600 python tools/bench_alt.py --workload ppo --steps 256 2>&1 | tail -1 > $O/ppo_$v.json
  python -c "
import json; d=json.load(open('$O/ppo_$v.json')); print('fold waves=$v  %.4e env-steps/s  %.2f ms per iteration  frac %.3f' % (d['value'], d['s_per_iteration']*1e3, d['roofline']['frac']))" | tee -a $O/ab.txt
done
