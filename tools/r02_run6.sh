#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r02f
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" >> $O/status
timeout 120 python tools/exp_stagger.py 65536 single > $O/times.log 2>&1
VISFLY_AMD_SPLIT=1 timeout 120 python tools/exp_stagger.py 65536 single >> $O/times.log 2>&1
timeout 120 python tools/exp_stagger.py 32768 single >> $O/times.log 2>&1
timeout 120 python tools/exp_stagger.py 1048576 single >> $O/times.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.log 2>&1
echo "bench rc=$?" >> $O/status
grep -v amdgpu $O/times.log; tail -4 $O/pytest_all.log
