#!/bin/bash
# A/B of the unlikely-branch layout of k_env_step (helper test, episode-end blocks laid out behind the hot path) in both regimes, bench's own method too
O=$GRAFT_REPO_ROOT/gpurun_out/r04b34; mkdir -p $O; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
VF_ALT_LIB=$PWD/tools/tmp/libvf_nounlikely.so timeout 300 python tools/exp_env_quad.py 65536 2>&1 | grep QUAD | sed 's/^/inline   /' | tee -a $O/ab.txt
timeout 300 python tools/exp_env_quad.py 65536 2>&1 | grep QUAD | sed 's/^/unlikely /' | tee -a $O/ab.txt
done
for v in inline unlikely inline unlikely; do
  L=""; [ $v = inline ] && L=$PWD/tools/tmp/libvf_nounlikely.so
  VF_ALT_LIB=$L timeout 600 python tools/bench_alt.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>&1 | grep "^{" > $O/bench_$v.json
  python -c "
import json; j=json.load(open('$O/bench_$v.json')); print('$v  value %.4e kernel %.2f | resets %.4e kernel %.2f x%.3f' % (j['value'], j['roofline']['kernel_us'], j['with_resets']['value'], j['with_resets']['kernel_us'], j['with_resets']['kernel_us_vs_headline']))" | tee -a $O/ab.txt
done
