#!/bin/bash
# MFMA-pipe occupancy of the register-chained kernels (r03): SQ_VALU_MFMA_BUSY_CYCLES (cycles) against SQ_BUSY_CYCLES / SQ_WAVE_CYCLES
# (quad-cycles) per dispatch, PMC pass without trace domains
O=$GRAFT_REPO_ROOT/gpurun_out/r03m
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
RP="rocprofv3 --output-format csv --kernel-trace"
timeout 600 $RP --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/mf_ppo -- python $R/bench.py --workload ppo --steps 256 > $O/log_ppo.txt 2>&1
timeout 600 $RP --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/mf_bptt -- python $R/bench.py --workload bptt --steps 128 > $O/log_bptt.txt 2>&1
for k in k_ppo_update_chain k_ppo_rollout k_mlp_wgrad; do echo "== $k"; python $R/tools/pmc_summary.py /tmp/mf_ppo $k; done > $O/pmc_mfma.txt 2>&1
for k in k_bptt_rollout k_bptt_reverse k_mlp_wgrad; do echo "== $k (bptt)"; python $R/tools/pmc_summary.py /tmp/mf_bptt $k; done >> $O/pmc_mfma.txt 2>&1
cat $O/pmc_mfma.txt
