#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r02s
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/p_bptt -- python $R/bench.py --workload bptt --steps 128 > $O/bench_bptt_profiled.log 2>&1
python $R/tools/prof_summary.py $(ls /tmp/p_bptt/*/*kernel_stats.csv | head -1) $O/bptt_kernel_stats.txt "python bench.py --workload bptt --steps 128" > /dev/null 2>&1
head -12 $O/bptt_kernel_stats.txt
