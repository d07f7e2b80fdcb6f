#!/bin/bash
# final env-step evidence on the shipped build (preloaded arguments + SLP threshold 8): rocprofv3 kernel stats, traffic and SQ counters
O=$GRAFT_REPO_ROOT/gpurun_out/r04b37; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/p_env -- python $R/bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --no-reset-leg --sustain-s 1 > $O/bench_env_profiled.log 2>&1
python $R/tools/prof_summary.py $(ls /tmp/p_env/*/*kernel_stats.csv | head -1) $O/r04_env_step_kernel_stats.txt "python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --no-reset-leg --sustain-s 1" | head -6
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d /tmp/pm_f -- python $R/tools/exp_env_one.py 65536 12 > $O/log_f.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d /tmp/pm_w -- python $R/tools/exp_env_one.py 65536 12 > $O/log_w.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pm_a -- python $R/tools/exp_env_one.py 65536 12 > $O/log_a.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -d /tmp/pm_b -- python $R/tools/exp_env_one.py 65536 12 > $O/log_b.txt 2>&1
for p in f w a b; do python $R/tools/pmc_summary.py /tmp/pm_$p k_env_step >> $O/pmc_env.txt 2>&1; done; cat $O/pmc_env.txt
