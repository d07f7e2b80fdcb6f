#!/bin/bash
# r04 batch 13: kernel stats of the SHAC and reference-actor BPTT iterations after the persistent launches
O=$GRAFT_REPO_ROOT/gpurun_out/r04b13; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
RP="rocprofv3 --output-format csv"
timeout 600 $RP --kernel-trace --stats -d /tmp/p_shac -- python $R/bench.py --workload shac --steps 256 > $O/bench_shac_profiled.log 2>&1
python $R/tools/prof_summary.py $(ls /tmp/p_shac/*/*kernel_stats.csv | head -1) $O/r04_shac_kernel_stats.txt "python bench.py --workload shac --steps 256" > /dev/null 2>&1
head -30 $O/r04_shac_kernel_stats.txt
