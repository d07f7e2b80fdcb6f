#!/bin/bash
# r04 final evidence: full GPU suite, the driver's bench command, rocprofv3 kernel stats of the four workloads, PMC passes (env SQ + MFMA)
O=$GRAFT_REPO_ROOT/gpurun_out/r04f; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; tail -3 $O/pytest_all.txt | head -2
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu > $O/bench_default.txt
timeout 900 python bench.py 2>&1 | grep -v amdgpu > $O/bench_2000.txt
python - <<'PY'
import json,glob,os
for n in ("bench_default","bench_2000"):
    for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04f/%s.txt'%n):
        if l.startswith('{'):
            j=json.loads(l); print(n, 'value %.4e'%j['value'], 'ms', j['ms_per_step'], 'frac', round(j['roofline']['frac'],3), 'kernel_us', j['roofline'].get('kernel_us'), 'sustained %.3e'%j['sustained']['value'], 'resets %.3e x%.3f'%(j['with_resets']['value'], j['with_resets']['kernel_us_vs_headline']), 'cpu %.3e'%j['cpu_baseline']['value'])
            for k,v in j.get('secondary',{}).items(): print('   ', k, '%.4e'%v['value'], round(v['roofline']['frac'],3))
PY
bash tools/r04_profiles.sh > $O/profiles.log 2>&1; tail -40 $O/profiles.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pm_a -- python $R/tools/exp_env_one.py 65536 12 > $O/log_a.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -d /tmp/pm_b -- python $R/tools/exp_env_one.py 65536 12 > $O/log_b.txt 2>&1
for p in a b; do python $R/tools/pmc_summary.py /tmp/pm_$p k_env_step >> $O/pmc_env_sq.txt 2>&1; done; cat $O/pmc_env_sq.txt
for w in ppo bptt; do
  steps=256; [ $w = bptt ] && steps=128
  timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pmm_$w -- python $R/bench.py --workload $w --steps $steps > $O/log_mfma_$w.txt 2>&1
  for k in k_ppo_update_chain k_ppo_rollout k_mlp_wgrad k_bptt_rollout k_bptt_reverse; do echo "== $k ($w)" >> $O/pmc_mfma.txt; python $R/tools/pmc_summary.py /tmp/pmm_$w $k >> $O/pmc_mfma.txt 2>&1; done
done
cat $O/pmc_mfma.txt
