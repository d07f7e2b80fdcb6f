#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r02k
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" >> $O/status
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/status
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.log 2>&1
echo "bench rc=$?" >> $O/status
timeout 600 python bench.py > $O/bench_default.log 2>&1
echo "bench_default rc=$?" >> $O/status
tail -3 $O/pytest_all.log; cat $O/status; tail -1 $O/bench_20.log | cut -c1-600
