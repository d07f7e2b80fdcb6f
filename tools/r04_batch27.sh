#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04b27; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/p_shac -- python $GRAFT_REPO_ROOT/bench.py --workload shac --steps 256 > $O/bench_shac.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $(ls /tmp/p_shac/*/*kernel_stats.csv | head -1) $O/r04_shac_kernel_stats.txt "python bench.py --workload shac --steps 256" | head -16
cd $GRAFT_REPO_ROOT; timeout 600 python -m pytest tests/test_shac_gpu.py -x -q -rs 2>&1 | tail -4
