#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r02o
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/p_ppo -- python $R/bench.py --workload ppo --steps 256 > $O/bench_ppo_profiled.log 2>&1
python $R/tools/prof_summary.py $(ls /tmp/p_ppo/*/*kernel_stats.csv | head -1) $O/ppo_kernel_stats.txt "python bench.py --workload ppo --steps 256" > /dev/null 2>&1
head -30 $O/ppo_kernel_stats.txt
