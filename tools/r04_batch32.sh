#!/bin/bash
# k_twin_q_update_chain keeping the ReLU masks as bits (16 per tile) and capped at 256 VGPRs (two waves per SIMD) vs uncapped (370 VGPRs, one wave): SHAC A/B + tests
O=$GRAFT_REPO_ROOT/gpurun_out/r04b32; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_shac_gpu.py -x -q 2>&1 | grep -E "passed|failed|error" | tee $O/pytest.txt
for v in 1 2 1 2; do
  L=""; [ $v = 1 ] && L=$PWD/tools/tmp/libvf_twinq1.so
  VF_ALT_LIB=$L timeout 600 python tools/bench_alt.py --workload shac --steps 256 2>&1 | tail -1 > $O/shac_$v.json
  python -c "
import json; d=json.load(open('$O/shac_$v.json')); print('twin-q waves per SIMD=$v  %.4e env-steps/s  %.3f ms per iteration  frac %.3f' % (d['value'], d['s_per_iteration']*1e3, d['roofline']['frac']))" | tee -a $O/ab.txt
done
