#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04b33; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pmm_shac -- python $R/bench.py --workload shac --steps 256 > $O/log_mfma_shac.txt 2>&1
for k in k_twin_q_update_chain k_mlp_wgrad k_mlp_forward_chain k_bptt_rollout k_bptt_reverse; do echo "== $k (shac)" >> $O/pmc_mfma_shac.txt; python $R/tools/pmc_summary.py /tmp/pmm_shac $k >> $O/pmc_mfma_shac.txt 2>&1; done
timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d /tmp/pmf_shac -- python $R/bench.py --workload shac --steps 256 > $O/log_f_shac.txt 2>&1
timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d /tmp/pmw_shac -- python $R/bench.py --workload shac --steps 256 > $O/log_w_shac.txt 2>&1
for k in k_twin_q_update_chain k_mlp_wgrad; do echo "== $k traffic" >> $O/pmc_mfma_shac.txt; python $R/tools/pmc_summary.py /tmp/pmf_shac $k >> $O/pmc_mfma_shac.txt; python $R/tools/pmc_summary.py /tmp/pmw_shac $k >> $O/pmc_mfma_shac.txt; done
cat $O/pmc_mfma_shac.txt
