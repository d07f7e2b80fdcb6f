#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r02i
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" >> $O/status
timeout 300 python tools/exp_fused.py > $O/fused.log 2>&1
timeout 120 python tools/exp_stagger.py 65536 single >> $O/fused.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.log 2>&1
echo "bench rc=$?" >> $O/status
timeout 600 python bench.py > $O/bench_default.log 2>&1
echo "bench_default rc=$?" >> $O/status
grep -v amdgpu $O/fused.log; tail -3 $O/pytest_all.log; cat $O/status
