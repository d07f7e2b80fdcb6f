#!/bin/bash
# r04 GPU batch 3: persistent launches with RK4 + drag DR (tests + timing), SQ counters of the new k_mlp_wgrad
O=$GRAFT_REPO_ROOT/gpurun_out/r04b3; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_ppo_gpu.py tests/test_bptt_gpu.py tests/test_dyn_gpu.py tests/test_env_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
timeout 600 python tools/exp_rk4_persistent.py 2>&1 | grep -v amdgpu | tee $O/rk4_persistent.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pm1 -- python $R/tools/exp_ppo_update_one.py 25600 20 > $O/log1.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD -d /tmp/pm2 -- python $R/tools/exp_ppo_update_one.py 25600 20 > $O/log2.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/pm3 -- python $R/tools/exp_ppo_update_one.py 25600 200 > $O/log3.txt 2>&1
for p in 1 2; do python $R/tools/pmc_summary.py /tmp/pm$p k_mlp_wgrad >> $O/pmc_wgrad.txt 2>&1; python $R/tools/pmc_summary.py /tmp/pm$p k_ppo_update_chain >> $O/pmc_chain.txt 2>&1; done
python $R/tools/prof_summary.py $(ls /tmp/pm3/*/*kernel_stats.csv | head -1) $O/update_one_kernel_stats.txt "python tools/exp_ppo_update_one.py 25600 200" > /dev/null 2>&1
cat $O/pmc_wgrad.txt; head -8 $O/update_one_kernel_stats.txt
