#!/bin/bash
# The gpurun sessions of round 5, one case per session:  gpurun -- 'bash tools/r05_sessions.sh <name>'
# Output under gpurun_out/r05_<name>/ ; the summaries that are cited go to profiles/r05_*.
S=$1; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_$S; mkdir -p $O; cd $R
export TMPDIR=/tmp
prof() {   # prof <tag> <cmd...>: rocprofv3 kernel stats of a command -> $O/<tag>_stats.txt
    local tag=$1; shift
    (cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_$tag -- "$@" > $O/prof_$tag.log 2>&1)
    python $R/tools/prof_summary.py $(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_stats.txt "$@" > /dev/null 2>&1
    head -12 $O/${tag}_stats.txt
}
case $S in
split1)   # branch-parallel fused update kernels: parity + A/B against the one-wave kernels
    timeout 900 python -m pytest tests/test_ppo_gpu.py tests/test_shac_gpu.py tests/test_ppo_golden.py -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt
    for sp in 0 1; do
        VISFLY_AMD_CHAIN_SPLIT=$sp prof ppo_split$sp timeout 300 python $R/tools/exp_ppo_update_one.py 25600 300
    done
    for sp in 0 1; do
        VISFLY_AMD_CHAIN_SPLIT=$sp timeout 600 python bench.py --workload ppo --no-cpu-baseline 2>&1 | tail -1 > $O/bench_ppo_split$sp.json
        VISFLY_AMD_CHAIN_SPLIT=$sp timeout 600 python bench.py --workload shac --no-cpu-baseline 2>&1 | tail -1 > $O/bench_shac_split$sp.json
    done
    python - <<PY
import json
for w in ("ppo", "shac"):
    for sp in (0, 1):
        try:
            d = json.loads(open("$O/bench_%s_split%d.json" % (w, sp)).read())
            print(w, "split", sp, d.get("value"), d.get("ms_per_step"), d.get("roofline", {}).get("frac"))
        except Exception as e:
            print(w, sp, "unparsed", e)
PY
    ;;
split2)   # where do the split kernels differ from the one-wave kernels?  + kernel stats
    timeout 600 python tools/tmp/dbg_split.py 2>&1 | tee $O/dbg.txt
    for sp in 0 1; do
        VISFLY_AMD_CHAIN_SPLIT=$sp prof ppo_split$sp timeout 300 python $R/tools/exp_ppo_update_one.py 25600 300
    done
    ;;
split3)   # split vs one-wave fused PPO step over the row count (1 / 2 / 4+ waves per SIMD): where does the second wave pay?
    timeout 600 python tools/tmp/dbg_split.py 2>&1 | grep -v amdgpu.ids | tee $O/dbg.txt
    for M in 8192 16384 25600 32768 65536 131072; do
        for sp in 0 1; do
            VISFLY_AMD_CHAIN_SPLIT=$sp prof m${M}_split$sp timeout 300 python $R/tools/exp_ppo_update_one.py $M 100 > /dev/null
            echo "M=$M split=$sp $(grep k_ppo_update $O/m${M}_split${sp}_stats.txt | awk '{print $(NF-4), $(NF-3), $(NF-2)}')" | tee -a $O/scan.txt
        done
    done
    ;;
split4)   # where do the waves of the split kernel land, and how long do they live?  (tools/tmp/libvf_strace.so = -DVF_SPLIT_TRACE build)
    for M in ${MS:-8192 16384 25600 32768}; do
        VF_ALT_LIB=$R/tools/tmp/libvf_strace.so timeout 300 python tools/exp_split_trace.py $M 2>&1 | grep -v amdgpu.ids | tee -a $O/trace.txt
    done
    ;;
pmc1)     # SQ counters of the fused PPO step, split vs one-wave, 25 600 rows: what do the waves wait for?
    cd /tmp
    for sp in 0 1; do
      i=0
      for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
                 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
                 "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
                 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_INSTS_SALU" \
                 "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM" \
                 "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS"; do
        i=$((i+1))
        VISFLY_AMD_CHAIN_SPLIT=$sp timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc $set -d /tmp/pmc_${sp}_$i -- python $R/tools/exp_ppo_update_one.py ${M:-25600} 20 > $O/log_${sp}_$i.txt 2>&1
        python $R/tools/pmc_summary.py /tmp/pmc_${sp}_$i k_ppo_update >> $O/pmc_split$sp.txt
      done
      echo "== split=$sp"; cat $O/pmc_split$sp.txt
    done
    ;;
abl1)     # ablations of k_ppo_update_split (trace builds): stores off, loss off, ring depth 4 -- which of them is the per-CU bottleneck?
    for v in ${VS:-base nostore noloss d4 both}; do
        echo "### variant $v" | tee -a $O/abl.txt
        for M in 8192 25600 32768; do
            VF_ALT_LIB=$R/tools/tmp/libvf_s_$v.so timeout 300 python tools/exp_split_trace.py $M 2>&1 | grep -v amdgpu.ids | grep -v "late wave" | tee -a $O/abl.txt
        done
    done
    ;;
occ)      # dependent fp32 MFMA chains vs waves per CU / per SIMD (tools/mfma_occupancy_probe.hip)
    timeout 300 tools/mfma_occupancy_probe 2>&1 | tee $O/occ.txt
    ;;
bufld)    # weight fragments as buffer loads: one-wave fused PPO step, shipped build vs the same kernel with the global loads (VF_ALT_LIB); + probe with stores
    timeout 300 tools/mfma_occupancy_probe 2>&1 | tee $O/occ.txt
    timeout 600 python -m pytest tests/test_ppo_gpu.py -x -q -m gpu -k "fused or chain" 2>&1 | tail -4 | tee $O/pytest.txt
    for rep in 1 2; do
    for lib in shipped globalloads; do
        alt=""; [ $lib = globalloads ] && alt=$R/tools/tmp/libvf_globalloads.so
        VF_ALT_LIB=$alt VISFLY_AMD_CHAIN_SPLIT=0 prof one_$lib timeout 300 python $R/tools/exp_ppo_update_one.py 25600 300 > /dev/null
        echo "rep $rep one-wave $lib: $(grep k_ppo_update $O/one_${lib}_stats.txt | awk '{print $(NF-4), $(NF-3), $(NF-2)}')" | tee -a $O/ab.txt
    done
    done
    for M in 8192 16384 25600; do for sp in 0 1; do
        VISFLY_AMD_CHAIN_SPLIT=$sp prof m${M}_$sp timeout 300 python $R/tools/exp_ppo_update_one.py $M 200 > /dev/null
        echo "M=$M split=$sp: $(grep k_ppo_update $O/m${M}_${sp}_stats.txt | awk '{print $(NF-4), $(NF-3), $(NF-2)}')" | tee -a $O/ab.txt
    done; done
    ;;
full)     # the whole GPU suite on the shipped build + bench legs
    timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest.txt
    timeout 300 tools/mfma_occupancy_probe 2>&1 | tee $O/occ.txt
    for M in 8192 16384 25600; do for sp in 0 1; do
        VISFLY_AMD_CHAIN_SPLIT=$sp prof m${M}_$sp timeout 300 python $R/tools/exp_ppo_update_one.py $M 200 > /dev/null
        echo "M=$M split=$sp: $(grep k_ppo_update $O/m${M}_${sp}_stats.txt | awk '{print $(NF-4), $(NF-3), $(NF-2)}')" | tee -a $O/ab.txt
    done; done
    for sp in 0 1; do
        VISFLY_AMD_CHAIN_SPLIT=$sp timeout 600 python bench.py --workload ppo --no-cpu-baseline 2>&1 | tail -1 > $O/bench_ppo_split$sp.json
        VISFLY_AMD_CHAIN_SPLIT=$sp timeout 600 python bench.py --workload shac --no-cpu-baseline 2>&1 | tail -1 > $O/bench_shac_split$sp.json
    done
    timeout 600 python bench.py --workload bptt --no-cpu-baseline 2>&1 | tail -1 > $O/bench_bptt.json
    python - <<PY
import json
for w in ("ppo", "shac"):
    for sp in (0, 1):
        try:
            d = json.loads(open("$O/bench_%s_split%d.json" % (w, sp)).read())
            print(w, "split", sp, d.get("value"), d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("us_per_update"))
        except Exception as e:
            print(w, sp, "unparsed", e)
d = json.loads(open("$O/bench_bptt.json").read()); print("bptt", d.get("value"), d.get("reference_actor", {}))
PY
    ;;
full2)    # whole suite + threshold scan of the split form + the three trainer benches
    timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.txt
    for M in 20480 25600 32768 49152; do for sp in 0 1; do
        VISFLY_AMD_CHAIN_SPLIT=$sp prof m${M}_$sp timeout 300 python $R/tools/exp_ppo_update_one.py $M 200 > /dev/null
        echo "M=$M split=$sp: $(grep k_ppo_update $O/m${M}_${sp}_stats.txt | awk '{print $(NF-4), $(NF-3), $(NF-2)}')" | tee -a $O/ab.txt
    done; done
    for w in ppo bptt shac; do
        timeout 600 python bench.py --workload $w --no-cpu-baseline 2>&1 | tail -1 > $O/bench_$w.json
    done
    python - <<PY
import json
for w in ("ppo", "bptt", "shac"):
    d = json.loads(open("$O/bench_%s.json" % w).read())
    print(w, d.get("value"), d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("us_per_update"), d.get("reference_actor", {}).get("value"), d.get("split_ms"))
PY
    ;;
evidence) # r05 evidence for profiles/: the driver's bench command, kernel stats of every bench workload, PMC passes (env traffic + SQ, MFMA of the trainers)
    timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_default.json
    timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench_2000steps.json
    cd /tmp
    RP="rocprofv3 --output-format csv"
    timeout 600 $RP --kernel-trace --stats -d /tmp/ks_env -- python $R/bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --no-reset-leg --sustain-s 1 > $O/log_ks_env.txt 2>&1
    python $R/tools/prof_summary.py $(find /tmp/ks_env -name "*kernel_stats.csv" | head -1) $O/r05_env_step_kernel_stats.txt python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --no-reset-leg --sustain-s 1 > /dev/null
    for w in ppo bptt shac nav_rk4_dr; do
        steps=256; [ $w = bptt ] && steps=128; [ $w = nav_rk4_dr ] && steps=200
        timeout 600 $RP --kernel-trace --stats -d /tmp/ks_$w -- python $R/bench.py --workload $w --steps $steps --no-cpu-baseline > $O/log_ks_$w.txt 2>&1
        python $R/tools/prof_summary.py $(find /tmp/ks_$w -name "*kernel_stats.csv" | head -1) $O/r05_${w}_kernel_stats.txt python bench.py --workload $w --steps $steps > /dev/null
    done
    timeout 300 $RP --kernel-trace --pmc FETCH_SIZE -d /tmp/pm_f -- python $R/tools/exp_env_one.py 65536 12 > $O/log_f.txt 2>&1
    timeout 300 $RP --kernel-trace --pmc WRITE_SIZE -d /tmp/pm_w -- python $R/tools/exp_env_one.py 65536 12 > $O/log_w.txt 2>&1
    for p in f w; do python $R/tools/pmc_summary.py /tmp/pm_$p k_env_step >> $O/pmc_traffic.txt 2>&1; done
    timeout 300 $RP --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pm_a -- python $R/tools/exp_env_one.py 65536 12 > $O/log_a.txt 2>&1
    timeout 300 $RP --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -d /tmp/pm_b -- python $R/tools/exp_env_one.py 65536 12 > $O/log_b.txt 2>&1
    for p in a b; do python $R/tools/pmc_summary.py /tmp/pm_$p k_env_step >> $O/pmc_env_sq.txt 2>&1; done
    for w in ppo bptt shac; do
        steps=256; [ $w = bptt ] && steps=128
        timeout 600 $RP --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pmm_$w -- python $R/bench.py --workload $w --steps $steps --no-cpu-baseline > $O/log_mfma_$w.txt 2>&1
        for k in k_ppo_update_split k_ppo_update_chain k_ppo_rollout k_mlp_wgrad k_bptt_rollout k_bptt_reverse k_twin_q_update; do echo "== $k ($w)" >> $O/pmc_mfma.txt; python $R/tools/pmc_summary.py /tmp/pmm_$w $k >> $O/pmc_mfma.txt 2>&1; done
    done
    cat $O/pmc_traffic.txt $O/pmc_env_sq.txt; head -8 $O/r05_*_kernel_stats.txt
    ;;
prefetch) # next epoch's gather + advantage normalisation on a side stream under the optimiser steps
    timeout 600 python -m pytest tests/test_ppo_gpu.py tests/test_ppo_loop_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest.txt
    timeout 900 python tools/exp_ppo_prefetch.py 6 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
    (cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace -d /tmp/prof_pf -- python $R/tools/exp_ppo_prefetch.py 1 only > $O/prof.log 2>&1)
    python tools/trace_overlap.py /tmp/prof_pf k_gather_rows | tee $O/overlap.txt
    ;;
jit)      # chain kernels compiled on first use for any net_arch -- parity, then the optimiser step with / without the plugin
    timeout 1200 python -m pytest tests/test_chain_jit_gpu.py -x -q 2>&1 | tail -5 | tee $O/pytest.txt
    timeout 900 python tools/exp_chain_jit.py 2>&1 | grep -v amdgpu.ids | tee $O/table.txt
    ;;
functional) # end of the round: soak of the step kernels, PPO learning curve (built-in class) and on a generated class
    timeout 1500 python tools/soak_envs.py 50000 2>&1 | grep -v amdgpu.ids | tee $O/soak.txt
    timeout 900 python tools/exp_ppo_learn.py 40 2>&1 | grep -v amdgpu.ids | tee $O/ppo_learning_curve.txt
    timeout 900 python tools/exp_ppo_learn.py 40 pi=128,128:vf=32 2>&1 | grep -v amdgpu.ids | tee $O/ppo_learning_curve_generated.txt
    ;;
avail)    # counter names this rocprofv3 knows on gfx950
    (cd /tmp && rocprofv3 --list-avail > $O/avail.txt 2>&1); grep -c . $O/avail.txt
    ;;
*) echo "unknown session $S"; exit 2;;
esac
