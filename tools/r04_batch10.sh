#!/bin/bash
# r04 GPU batch 10: prefetched re-spawn without the headline's extra bytes (slot loads only in ending waves, stale bits for the helper blocks):
# parity tests, timing of both regimes, PMC traffic of k_env_step (modes: stale bits on / off)
O=$GRAFT_REPO_ROOT/gpurun_out/r04b10; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_env_gpu.py tests/test_env_multistep_gpu.py tests/test_env_external_scene_gpu.py tests/test_config_scale_gpu.py tests/test_ppo_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for b in 1 0; do VISFLY_AMD_STALE_BITS=$b timeout 300 python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --sustain-s 1 2>&1 | grep -v amdgpu > $O/bench_bits$b.txt; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04b10/bench_bits*.txt')):
    for l in open(f):
        if l.startswith('{'):
            j=json.loads(l); t=j['timing']; w=j['with_resets']; print(os.path.basename(f), 'value %.4e'%j['value'], 'event us', round(t['event_us_per_step'],2), '| resets %.4e'%w['value'], 'kernel', round(w['kernel_us'],2), 'ratio', round(w['kernel_us_vs_headline'],3))
PY
cd /tmp; export TMPDIR=/tmp
for b in 1 0; do
  export VISFLY_AMD_STALE_BITS=$b
  timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d /tmp/pm_f_$b -- python $R/tools/exp_env_one.py 65536 12 > $O/log_f_$b.txt 2>&1
  timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d /tmp/pm_w_$b -- python $R/tools/exp_env_one.py 65536 12 > $O/log_w_$b.txt 2>&1
  echo "== VISFLY_AMD_STALE_BITS=$b" >> $O/pmc_traffic.txt
  for p in f w; do python $R/tools/pmc_summary.py /tmp/pm_${p}_$b k_env_step >> $O/pmc_traffic.txt 2>&1; done
done
cat $O/pmc_traffic.txt
