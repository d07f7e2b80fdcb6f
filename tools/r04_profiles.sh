#!/bin/bash
# r04 evidence for profiles/: rocprofv3 kernel-trace stats of the four bench workloads (the commands the bench line's figures come from)
O=$GRAFT_REPO_ROOT/gpurun_out/r04f; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
RP="rocprofv3 --output-format csv"
timeout 600 $RP --kernel-trace --stats -d /tmp/p_env -- python $R/bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --no-reset-leg --sustain-s 1 > $O/bench_env_profiled.log 2>&1
python $R/tools/prof_summary.py $(ls /tmp/p_env/*/*kernel_stats.csv | head -1) $O/r04_env_step_kernel_stats.txt "python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --no-reset-leg --sustain-s 1" > /dev/null 2>&1
for w in ppo bptt shac; do
  steps=256; [ $w = bptt ] && steps=128
  timeout 600 $RP --kernel-trace --stats -d /tmp/p_$w -- python $R/bench.py --workload $w --steps $steps > $O/bench_${w}_profiled.log 2>&1
  python $R/tools/prof_summary.py $(ls /tmp/p_$w/*/*kernel_stats.csv | head -1) $O/r04_${w}_kernel_stats.txt "python bench.py --workload $w --steps $steps" > /dev/null 2>&1
done
head -8 $O/r04_env_step_kernel_stats.txt; head -12 $O/r04_ppo_kernel_stats.txt; head -12 $O/r04_bptt_kernel_stats.txt; head -24 $O/r04_shac_kernel_stats.txt
