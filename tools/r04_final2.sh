#!/bin/bash
# r04 final evidence after the preloaded-argument step kernels: kernel stats of the four workloads, PMC passes (env traffic + SQ, MFMA of the trainers), 2000-step bench
O=$GRAFT_REPO_ROOT/gpurun_out/r04f; mkdir -p $O; rm -f $O/pmc_*.txt; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python bench.py 2>&1 | grep -v amdgpu > $O/bench_2000.txt
bash tools/r04_profiles.sh > $O/profiles.log 2>&1; tail -60 $O/profiles.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d /tmp/pm_f -- python $R/tools/exp_env_one.py 65536 12 > $O/log_f.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d /tmp/pm_w -- python $R/tools/exp_env_one.py 65536 12 > $O/log_w.txt 2>&1
for p in f w; do python $R/tools/pmc_summary.py /tmp/pm_$p k_env_step >> $O/pmc_traffic.txt 2>&1; done; cat $O/pmc_traffic.txt
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pm_a -- python $R/tools/exp_env_one.py 65536 12 > $O/log_a.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -d /tmp/pm_b -- python $R/tools/exp_env_one.py 65536 12 > $O/log_b.txt 2>&1
for p in a b; do python $R/tools/pmc_summary.py /tmp/pm_$p k_env_step >> $O/pmc_env_sq.txt 2>&1; done; cat $O/pmc_env_sq.txt
for w in ppo bptt; do
  steps=256; [ $w = bptt ] && steps=128
  timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pmm_$w -- python $R/bench.py --workload $w --steps $steps > $O/log_mfma_$w.txt 2>&1
  for k in k_ppo_update_chain k_ppo_rollout k_mlp_wgrad k_bptt_rollout k_bptt_reverse; do echo "== $k ($w)" >> $O/pmc_mfma.txt; python $R/tools/pmc_summary.py /tmp/pmm_$w $k >> $O/pmc_mfma.txt 2>&1; done
done
cat $O/pmc_mfma.txt
