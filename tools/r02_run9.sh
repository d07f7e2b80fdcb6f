#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r02k
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" >> $O/status
timeout 600 python bench.py --workload ppo --steps 512 > $O/bench_ppo.log 2>&1
echo "ppo rc=$?" >> $O/status
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.log 2>&1
echo "bench rc=$?" >> $O/status
tail -3 $O/pytest_all.log; cat $O/status; grep "^{" $O/bench_ppo.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('PPO', d['value'], d['s_per_iteration'], d['split_ms'], d['roofline']['frac'])"
