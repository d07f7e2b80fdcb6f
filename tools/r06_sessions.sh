#!/bin/bash
# The gpurun sessions of round 6, one case per session:  gpurun -- 'bash tools/r06_sessions.sh <name>'
# Output under gpurun_out/r06_<name>/ ; the summaries that are cited go to profiles/r06_*.
S=$1; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_$S; mkdir -p $O; cd $R
export TMPDIR=/tmp
prof() {   # prof <tag> <cmd...>: rocprofv3 kernel stats of a command -> $O/<tag>_stats.txt
    local tag=$1; shift
    (cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_$tag -- "$@" > $O/prof_$tag.log 2>&1)
    python $R/tools/prof_summary.py $(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_stats.txt "$@" > /dev/null 2>&1
    head -14 $O/${tag}_stats.txt
}
ppo_ab() {  # ppo_ab <tag> <env assignments...>: bench.py --workload ppo under the given environment
    local tag=$1; shift
    env "$@" timeout 600 python bench.py --workload ppo --no-cpu-baseline 2>&1 | tail -1 > $O/bench_ppo_$tag.json
    python - <<PY
import json
try:
    d = json.loads(open("$O/bench_ppo_$tag.json").read())
    print("ppo $tag value %.4g  us/update %.2f  frac %.4f  train ms %.2f" % (d["value"], d["roofline"]["us_per_update"], d["roofline"]["frac"], d["split_ms"]["train"]))
except Exception as e:
    print("ppo $tag unparsed", e)
PY
}
case $S in
tail1)    # fused optimiser tail (vf_mlp_weight_grad_adam): parity with the separate launches, then A/B of the PPO loop and kernel stats
    timeout 900 python -m pytest tests/test_ppo_gpu.py -x -q -m gpu -k "fused_optimiser_tail or fused_tail or bitwise or fold_partials" 2>&1 | tail -15 | tee $O/pytest.txt
    timeout 900 python -m pytest tests/test_ppo_loop_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest_loop.txt
    for ft in 1 0; do ppo_ab tail$ft VISFLY_AMD_FUSED_TAIL=$ft | tee -a $O/ab.txt; done
    for ft in 1 0; do
        VISFLY_AMD_FUSED_TAIL=$ft prof ppo_tail$ft timeout 300 python $R/bench.py --workload ppo --no-cpu-baseline --steps 256
    done
    ;;
tail2)    # wave timeline of the fused launch (trace build) + the A/B again
    timeout 600 python -m pytest tests/test_ppo_gpu.py -x -q -m gpu -k "fused_optimiser_tail or fused_tail" 2>&1 | tail -5 | tee $O/pytest.txt
    VF_ALT_LIB=$R/tools/tmp/libvf_wtrace.so timeout 300 python tools/exp_tail_trace.py 25600 2>&1 | grep -v amdgpu.ids | tee $O/trace.txt
    for ft in 1 0 1 0; do ppo_ab tail$ft VISFLY_AMD_FUSED_TAIL=$ft | tee -a $O/ab.txt; done
    for ft in 1 0; do
        VISFLY_AMD_FUSED_TAIL=$ft prof ppo_tail$ft timeout 300 python $R/bench.py --workload ppo --no-cpu-baseline --steps 256
    done
    ;;
aux)      # cache policy of the chain kernel's saved-activation stores (write-through / non-temporal): does the chain -> wgrad boundary shrink?
    ppo_ab base VISFLY_AMD_FUSED_TAIL=0 | tee -a $O/ab.txt
    for a in 16 2 18; do
        VF_ALT_LIB=$R/tools/tmp/libvf_saux$a.so timeout 600 python tools/bench_alt.py --workload ppo --no-cpu-baseline 2>&1 | tail -1 > $O/bench_ppo_aux$a.json
        python - <<PY | tee -a $O/ab.txt
import json
d = json.loads(open("$O/bench_ppo_aux$a.json").read())
print("ppo store aux $a value %.4g  us/update %.2f  frac %.4f  train ms %.2f" % (d["value"], d["roofline"]["us_per_update"], d["roofline"]["frac"], d["split_ms"]["train"]))
PY
    done
    ppo_ab base VISFLY_AMD_FUSED_TAIL=0 | tee -a $O/ab.txt
    prof ppo_base timeout 300 python $R/bench.py --workload ppo --no-cpu-baseline --steps 256
    ;;
buckets)  # two-bucket gradient exchange: lockstep + equality tests (2 processes / 1 GPU over gloo), the N = 1 cost of the split launches
    timeout 1200 python -m pytest tests/test_parallel_gpu.py -x -q -m gpu 2>&1 | tail -40 | tee $O/pytest.txt
    VISFLY_AMD_GRAD_BUCKETS_FORCE=1 timeout 600 python bench.py --workload ppo --no-cpu-baseline 2>&1 | tail -1 > $O/bench_ppo_force.json
    python - <<PY | tee $O/ab.txt
import json
d = json.loads(open("$O/bench_ppo_force.json").read())
print("N = 1, split forced:", d["exchange"])
PY
    VISFLY_AMD_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --workload ppo --no-cpu-baseline --agents 8192 2>&1 | tail -1 > $O/bench_ppo_2ranks_gloo.json
    python - <<PY | tee -a $O/ab.txt
import json
d = json.loads(open("$O/bench_ppo_2ranks_gloo.json").read())
print("2 ranks on one GPU over gloo (control flow only):", d["value"], d["exchange"])
PY
    ;;
acts)     # optimiser-step fraction per activation (generated chain classes), + where the suite's time goes
    for a in relu tanh elu leaky_relu; do
        timeout 900 python bench.py --workload ppo --no-cpu-baseline --activation $a 2>&1 | tail -1 > $O/bench_ppo_$a.json
        python - <<PY | tee -a $O/acts.txt
import json
d = json.loads(open("$O/bench_ppo_$a.json").read())
print("activation_fn %-10s PPO %.4g env-steps/s  optimiser step %.2f us  frac %.4f  rollout ms %s" % ("$a", d["value"], d["roofline"]["us_per_update"], d["roofline"]["frac"], d["split_ms"]["rollout_256_steps"]))
PY
    done
    timeout 3000 python -m pytest tests -x -q -m gpu --durations=12 2>&1 | tail -30 | tee $O/pytest.txt
    ;;
bptt)     # (a) BPTT's weight-gradient launch per chunk of steps, cache-hot vs the one launch over the horizon; (b) why the driver's
          # reference-actor figure fell 3.03e8 (r04) -> 2.93e8 (r05): the three trees' own bench.py --workload bptt on THIS box, interleaved
    timeout 900 python tools/exp_bptt_wgrad_chunks.py 2>&1 | grep -v amdgpu.ids | tee $O/wgrad_chunks.txt
    for rep in 1 2 3; do
        for tree in tools/tmp/tree_13cc83c tools/tmp/tree_8743b9a .; do
            (cd $R/$tree && timeout 600 python bench.py --workload bptt --no-cpu-baseline 2>/dev/null | tail -1) > $O/bptt_$(basename $tree)_$rep.json
            python - <<PY | tee -a $O/ab.txt
import json
d = json.loads(open("$O/bptt_$(basename $tree)_$rep.json").read())
ra = d.get("reference_actor", {}).get("value") or d["value"]
mlp = d.get("mlp_policy_actor", {}).get("value") or (d["value"] if "reference_actor" in d else None)
print("%-22s rep $rep  reference actor %.4g   MlpPolicy actor %s" % ("$(basename $tree)", ra, "%.4g" % mlp if mlp else "-"))
PY
        done
    done
    ;;
genhor)   # VERDICT r05 items 3b / 3c measured: generated twin-critic class + BPTT plugins on a non-default net_arch (one process per mode)
    for w in 128,128 32; do
        for m in all loop r05 blocktile; do
            timeout 900 python tools/exp_generated_horizons.py $m $w 2>&1 | grep -v amdgpu.ids | tee -a $O/table.txt
        done
    done
    ;;
evidence) # r06 evidence for profiles/: the driver's bench command, kernel stats of every bench workload, PMC passes (env traffic + SQ, MFMA of the trainers)
    timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_default.json
    timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench_2000steps.json
    cd /tmp
    RP="rocprofv3 --output-format csv"
    timeout 600 $RP --kernel-trace --stats -d /tmp/ks_env -- python $R/bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --no-reset-leg --sustain-s 1 > $O/log_ks_env.txt 2>&1
    python $R/tools/prof_summary.py $(find /tmp/ks_env -name "*kernel_stats.csv" | head -1) $O/r06_env_step_kernel_stats.txt python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --no-reset-leg --sustain-s 1 > /dev/null
    for w in ppo bptt shac nav_rk4_dr; do
        steps=256; [ $w = bptt ] && steps=128; [ $w = nav_rk4_dr ] && steps=200
        timeout 600 $RP --kernel-trace --stats -d /tmp/ks_$w -- python $R/bench.py --workload $w --steps $steps --no-cpu-baseline > $O/log_ks_$w.txt 2>&1
        python $R/tools/prof_summary.py $(find /tmp/ks_$w -name "*kernel_stats.csv" | head -1) $O/r06_${w}_kernel_stats.txt python bench.py --workload $w --steps $steps > /dev/null
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
    cat $O/pmc_traffic.txt $O/pmc_env_sq.txt; head -8 $O/r06_*_kernel_stats.txt
    ;;
functional) # end of the round: soak of the step kernels, PPO learning curve on the built-in class and on the reference's DEFAULT (Tanh) policy
    timeout 1500 python tools/soak_envs.py 50000 2>&1 | grep -v amdgpu.ids | tee $O/soak.txt
    timeout 900 python tools/exp_ppo_learn.py 40 2>&1 | grep -v amdgpu.ids | tee $O/ppo_learning_curve.txt
    timeout 900 python tools/exp_ppo_learn.py 40 default 2>&1 | grep -v amdgpu.ids | tee $O/ppo_learning_curve_tanh.txt
    timeout 900 python tools/exp_td_learn.py 60 2>&1 | grep -v amdgpu.ids | tee $O/td_learning_curves.txt
    ;;
tdlearn)  # BPTT / SHAC learning curves only
    timeout 900 python tools/exp_td_learn.py 60 2>&1 | grep -v amdgpu.ids | tee $O/td_learning_curves.txt
    ;;
fold)     # k_wgrad_fold with 8 / 16 / 32 / 64 partial rows in flight per lane (-DVF_FOLD_BATCH=1 / 2 / 4 / 8 builds): parity, A/B of the PPO step, kernel stats
    timeout 900 python -m pytest tests/test_ppo_gpu.py -x -q -m gpu -k "fused_optimiser_tail or fused_tail or bitwise or fold_partials or weight_grad" 2>&1 | tail -5 | tee $O/pytest.txt
    for rep in 1 2; do
        for b in 1 2 4 8; do
            lib=$R/tools/tmp/libvf_fold$b.so; [ $b = 4 ] && lib=$R/visfly_amd/csrc/libvisfly_amd.so
            VF_ALT_LIB=$lib timeout 600 python tools/bench_alt.py --workload ppo --no-cpu-baseline 2>&1 | tail -1 > $O/bench_ppo_fold$b.json
            python - <<PY | tee -a $O/ab.txt
import json
d = json.loads(open("$O/bench_ppo_fold$b.json").read())
print("fold batch $b rep $rep: value %.4g  us/update %.2f  frac %.4f  train ms %.2f" % (d["value"], d["roofline"]["us_per_update"], d["roofline"]["frac"], d["split_ms"]["train"]))
PY
        done
    done
    for b in 1 4; do
        lib=$R/tools/tmp/libvf_fold$b.so; [ $b = 4 ] && lib=$R/visfly_amd/csrc/libvisfly_amd.so
        VF_ALT_LIB=$lib prof ppo_fold$b timeout 300 python $R/tools/bench_alt.py --workload ppo --no-cpu-baseline --steps 256
    done
    ;;
tests)    # the whole GPU suite
    timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee $O/pytest.txt
    ;;
*) echo "unknown session $S"; exit 2 ;;
esac
