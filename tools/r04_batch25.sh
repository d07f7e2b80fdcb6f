#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04b25; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ppo_gpu.py -x -q 2>&1 | tail -2 | tee $O/pytest.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/p_ppo -- python $GRAFT_REPO_ROOT/bench.py --workload ppo --steps 256 > $O/bench_ppo.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $(ls /tmp/p_ppo/*/*kernel_stats.csv | head -1) $O/ppo_kernel_stats.txt "python bench.py --workload ppo --steps 256" | head -14
tail -1 $O/bench_ppo.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ppo %.4e'%d['value'], d['s_per_iteration'], d['roofline']['frac'])"
