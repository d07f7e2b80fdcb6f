#!/bin/bash
# r03 evidence for profiles/: kernel-trace stats of the bench workloads, PMC passes of the env-step kernel (with and without the
# helper blocks of the prefetched re-spawn), phase timings
O=$GRAFT_REPO_ROOT/gpurun_out/r03p
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
RP="rocprofv3 --output-format csv"
# 1. kernel stats of the same commands the bench line comes from
timeout 600 $RP --kernel-trace --stats -d /tmp/p_env -- python $R/bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --no-reset-leg > $O/bench_env_profiled.log 2>&1
python $R/tools/prof_summary.py $(ls /tmp/p_env/*/*kernel_stats.csv | head -1) $O/r03_env_step_kernel_stats.txt "python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --no-reset-leg" > /dev/null 2>&1
timeout 600 $RP --kernel-trace --stats -d /tmp/p_ppo -- python $R/bench.py --workload ppo --steps 256 > $O/bench_ppo_profiled.log 2>&1
python $R/tools/prof_summary.py $(ls /tmp/p_ppo/*/*kernel_stats.csv | head -1) $O/r03_ppo_kernel_stats.txt "python bench.py --workload ppo --steps 256" > /dev/null 2>&1
timeout 600 $RP --kernel-trace --stats -d /tmp/p_bptt -- python $R/bench.py --workload bptt --steps 128 > $O/bench_bptt_profiled.log 2>&1
python $R/tools/prof_summary.py $(ls /tmp/p_bptt/*/*kernel_stats.csv | head -1) $O/r03_bptt_kernel_stats.txt "python bench.py --workload bptt --steps 128" > /dev/null 2>&1
# 2. PMC: HBM traffic (separate passes) and SQ counters of the env-step kernel; mode 0 = prefetched re-spawn off (1024 main waves only,
#    comparable with r02), mode 3 = on (1024 main + 1024 helper waves)
for mode in 0 3; do
  export VISFLY_AMD_PREFETCH_MODE=$mode
  timeout 300 $RP --kernel-trace --pmc FETCH_SIZE -d /tmp/pm_f_$mode -- python $R/tools/exp_env_one.py 65536 12 > $O/log_f_$mode.txt 2>&1
  timeout 300 $RP --kernel-trace --pmc WRITE_SIZE -d /tmp/pm_w_$mode -- python $R/tools/exp_env_one.py 65536 12 > $O/log_w_$mode.txt 2>&1
  timeout 300 $RP --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pm_a_$mode -- python $R/tools/exp_env_one.py 65536 12 > $O/log_a_$mode.txt 2>&1
  timeout 300 $RP --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -d /tmp/pm_b_$mode -- python $R/tools/exp_env_one.py 65536 12 > $O/log_b_$mode.txt 2>&1
  for p in f w a b; do python $R/tools/pmc_summary.py /tmp/pm_${p}_$mode k_env_step >> $O/pmc_mode$mode.txt 2>&1; done
done
unset VISFLY_AMD_PREFETCH_MODE
cd $R
timeout 300 python tools/exp_bptt_phases.py 2>&1 | grep -v amdgpu > $O/bptt_phases.txt
timeout 300 python tools/exp_reset_bound.py 2>&1 | grep -v amdgpu > $O/reset_regime.txt
rm -f $O/log_*.txt
ls -la $O; cat $O/pmc_mode0.txt $O/pmc_mode3.txt; head -8 $O/r03_env_step_kernel_stats.txt; head -10 $O/r03_ppo_kernel_stats.txt; head -10 $O/r03_bptt_kernel_stats.txt
