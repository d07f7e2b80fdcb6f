#!/bin/bash
# The gpurun sessions of round 4 (37 batches + the end-of-round evidence runs), one case per session, in the order they ran:
#     gpurun -- 'bash tools/r04_sessions.sh <name>'        <name> = b1 .. b37 | final | final2 | profiles
# Folded from tools/r04_batchN.sh in round 5; what each session measured is the comment at its head, the
# file under profiles/ it fed is listed in tools/README.md.  Sessions that exercised experiment builds (-DVF_EXP_*) need the
# git revision of their round: those switches were removed from the sources in round 5.
case "$1" in
b1)
# r04 GPU batch 1: fused-weight-gradient skeleton, Infinity-Cache test of k_mlp_wgrad, two waves per SIMD in k_mlp_wgrad (A/B)
O=$GRAFT_REPO_ROOT/gpurun_out/r04b1; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 120 tools/wgrad_fused_probe > $O/wgrad_fused_probe.txt 2>&1
for w in 1 2; do VISFLY_AMD_WGRAD_WPS=$w timeout 200 python tools/exp_wgrad_mall.py 2>&1 | grep -v amdgpu > $O/wgrad_mall_wps$w.txt; done
for w in 1 2; do VISFLY_AMD_WGRAD_WPS=$w timeout 300 python bench.py --workload ppo --steps 256 2>&1 | grep -v amdgpu > $O/bench_ppo_wps$w.txt; done
VISFLY_AMD_WGRAD_WPS=2 timeout 900 python -m pytest tests/test_ppo_gpu.py tests/test_bptt_gpu.py tests/test_shac_gpu.py -x -q -m gpu > $O/pytest_wps2.txt 2>&1
tail -3 $O/pytest_wps2.txt; cat $O/wgrad_fused_probe.txt $O/wgrad_mall_wps*.txt; tail -c 1500 $O/bench_ppo_wps1.txt; echo; tail -c 1500 $O/bench_ppo_wps2.txt
    ;;
b2)
# r04 GPU batch 2: k_mlp_wgrad with slab-bounded buffer loads (A/B one vs two waves per SIMD) + the whole GPU suite on the 256-step fixtures
O=$GRAFT_REPO_ROOT/gpurun_out/r04b2; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
for w in 1 2; do VISFLY_AMD_WGRAD_WPS=$w timeout 200 python tools/exp_wgrad_mall.py 2>&1 | grep -v amdgpu > $O/wgrad_mall_wps$w.txt; done
for w in 1 2; do VISFLY_AMD_WGRAD_WPS=$w timeout 300 python bench.py --workload ppo --steps 256 2>&1 | grep -v amdgpu > $O/bench_ppo_wps$w.txt; done
timeout 300 python bench.py --workload bptt --steps 128 2>&1 | grep -v amdgpu > $O/bench_bptt.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
tail -5 $O/pytest_all.txt; cat $O/wgrad_mall_wps*.txt
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04b2/bench_*.txt')):
    for l in open(f):
        if l.startswith('{'):
            j=json.loads(l); print(os.path.basename(f), j['value'], j.get('split_ms'), j['roofline'].get('us_per_update'), j['roofline'].get('frac'))
PY
    ;;
b3)
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
    ;;
b4)
# r04 GPU batch 4: sub-step tape (k_bptt_rollout -> k_bptt_reverse), BPTT.learn loop fixture with the reference's actor
O=$GRAFT_REPO_ROOT/gpurun_out/r04b4; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_bptt_gpu.py tests/test_shac_gpu.py tests/test_abi.py tests/test_config_scale_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
for m in 0 1; do echo "== VISFLY_AMD_SUBSTEP_TAPE=$m" >> $O/bptt_phases.txt; VISFLY_AMD_SUBSTEP_TAPE=$m timeout 400 python tools/exp_bptt_phases.py 2>&1 | grep -v amdgpu >> $O/bptt_phases.txt; done
cat $O/bptt_phases.txt
for m in 0 1; do VISFLY_AMD_SUBSTEP_TAPE=$m timeout 300 python bench.py --workload bptt --steps 128 2>&1 | grep -v amdgpu > $O/bench_bptt_tape$m.txt; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04b4/bench_*.txt')):
    for l in open(f):
        if l.startswith('{'):
            j=json.loads(l); print(os.path.basename(f), j['value'], j['roofline'].get('frac'))
PY
    ;;
b5)
# r04 GPU batch 5: SAC-actor chain classes (tests, SHAC timing chain vs block-tile), full default bench line
O=$GRAFT_REPO_ROOT/gpurun_out/r04b5; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_ppo_gpu.py tests/test_shac_gpu.py tests/test_bptt_gpu.py -x -q -m gpu -k "sac or shac or loop or reference_actor or chain" > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
for m in 1 0; do VISFLY_AMD_MLP_CHAIN=$m timeout 300 python bench.py --workload shac --steps 256 2>&1 | grep -v amdgpu > $O/bench_shac_chain$m.txt; done
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu > $O/bench_default.txt
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04b5/bench_*.txt')):
    for l in open(f):
        if l.startswith('{'):
            j=json.loads(l); print(os.path.basename(f), j['value'], j.get('s_per_iteration'), j['roofline'].get('frac'))
            if 'secondary' in j:
                for k,v in j['secondary'].items(): print('   secondary', k, v and v.get('value'), v and v['roofline'].get('frac'))
PY
    ;;
b6)
# r04 GPU batch 6: k_mlp_wgrad prefetch-ring depth A/B (three builds), env-step launch time vs agents, SAC chain tests
O=$GRAFT_REPO_ROOT/gpurun_out/r04b6; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for lib in visfly_amd/csrc/libvisfly_amd.so tools/libvf_wg96.bin tools/libvf_wg128.bin; do echo "== $lib" >> $O/wgrad_depth.txt; timeout 200 python tools/exp_wgrad_mall.py 25600 30 $R/$lib 2>&1 | grep -v amdgpu | sed -n 2,3p >> $O/wgrad_depth.txt; done; done
cat $O/wgrad_depth.txt
timeout 300 python tools/exp_env_scaling.py 2>&1 | grep -v amdgpu | tee $O/env_scaling.txt
timeout 1500 python -m pytest tests/test_ppo_gpu.py tests/test_shac_gpu.py -x -q -m gpu -k "sac or shac" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
    ;;
b7)
# r04 GPU batch 7: completion-wait mode of the timed region (HSA_ENABLE_INTERRUPT) on the headline's --steps 20 regions; SAC tests
O=$GRAFT_REPO_ROOT/gpurun_out/r04b7; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for m in 1 0; do HSA_ENABLE_INTERRUPT=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --sustain-s 1 2>&1 | grep -v amdgpu > $O/bench_int$m.$rep.txt; done; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04b7/bench_int*.txt')):
    for l in open(f):
        if l.startswith('{'):
            j=json.loads(l); t=j['timing']; print(os.path.basename(f), 'value %.4e'%j['value'], 'ms/step', round(j['ms_per_step']*1e3,2), 'at completion', round(t['ms_per_step_at_completion']*1e3,2), 'event', round(t['event_us_per_step'],2), 'host', round(t['host_us_per_step'],2), 'resets %.4e'%j['with_resets']['value'])
PY
timeout 900 python -m pytest tests/test_ppo_gpu.py -x -q -m gpu -k "sac" 2>&1 | tail -3
    ;;
b8)
# r04 GPU batch 8: twin-critic chain class (tests + SHAC timing), adjoint division trim (BPTT tests + phases), full GPU suite
O=$GRAFT_REPO_ROOT/gpurun_out/r04b8; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_ppo_gpu.py -x -q -m gpu -k "critic or sac" > $O/pytest_critic.txt 2>&1; tail -15 $O/pytest_critic.txt
timeout 900 python -m pytest tests/test_shac_gpu.py tests/test_bptt_gpu.py -x -q -m gpu > $O/pytest_shac_bptt.txt 2>&1; tail -5 $O/pytest_shac_bptt.txt
for m in 1 0; do VISFLY_AMD_MLP_CHAIN=$m timeout 300 python bench.py --workload shac --steps 256 2>&1 | grep -v amdgpu > $O/bench_shac_chain$m.txt; done
timeout 300 python bench.py --workload bptt --steps 128 2>&1 | grep -v amdgpu > $O/bench_bptt.txt
timeout 400 python tools/exp_bptt_phases.py 2>&1 | grep -v amdgpu | head -8 > $O/bptt_phases.txt; cat $O/bptt_phases.txt
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04b8/bench_*.txt')):
    for l in open(f):
        if l.startswith('{'):
            j=json.loads(l); print(os.path.basename(f), j['value'], j.get('s_per_iteration'), j['roofline'].get('frac'))
PY
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; tail -5 $O/pytest_all.txt
    ;;
b9)
# r04 GPU batch 9: chain-kernel stores with per-layer pinned bases (no per-store scalar loads / exec branches): tests + timing of all three trainers
O=$GRAFT_REPO_ROOT/gpurun_out/r04b9; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 300 python bench.py --workload ppo --steps 256 2>&1 | grep -v amdgpu > $O/bench_ppo.txt
timeout 300 python bench.py --workload bptt --steps 128 2>&1 | grep -v amdgpu > $O/bench_bptt.txt
timeout 300 python bench.py --workload shac --steps 256 2>&1 | grep -v amdgpu > $O/bench_shac.txt
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04b9/bench_*.txt')):
    for l in open(f):
        if l.startswith('{'):
            j=json.loads(l); print(os.path.basename(f), j['value'], j.get('s_per_iteration'), j['roofline'].get('frac'), j['roofline'].get('us_per_update'))
PY
timeout 400 python tools/exp_bptt_phases.py 2>&1 | grep -v amdgpu | head -6
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/pm3 -- python $R/tools/exp_ppo_update_one.py 25600 200 > $O/log3.txt 2>&1
python $R/tools/prof_summary.py $(ls /tmp/pm3/*/*kernel_stats.csv | head -1) $O/update_one_kernel_stats.txt "python tools/exp_ppo_update_one.py 25600 200" > /dev/null 2>&1
head -7 $O/update_one_kernel_stats.txt
    ;;
b10)
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
    ;;
b11)
# r04 batch 11: persistent BPTT launches for the reference actor (td_policies.Actor) -- parity tests + timing
mkdir -p gpurun_out/r04b11
timeout 1200 python -m pytest tests/test_bptt_gpu.py tests/test_shac_gpu.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r04b11/pytest.txt
timeout 600 python tools/exp_bptt_refactor.py 2>&1 | tee gpurun_out/r04b11/ref_actor.txt
    ;;
b12)
# r04 batch 12: SHAC's horizon on the persistent launches -- parity + bench leg
mkdir -p gpurun_out/r04b12
timeout 1200 python -m pytest tests/test_bptt_gpu.py tests/test_shac_gpu.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r04b12/pytest.txt
timeout 600 python bench.py --workload shac --steps 256 2>&1 | tail -1 | tee gpurun_out/r04b12/bench_shac.json
    ;;
b13)
# r04 batch 13: kernel stats of the SHAC and reference-actor BPTT iterations after the persistent launches
O=$GRAFT_REPO_ROOT/gpurun_out/r04b13; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
RP="rocprofv3 --output-format csv"
timeout 600 $RP --kernel-trace --stats -d /tmp/p_shac -- python $R/bench.py --workload shac --steps 256 > $O/bench_shac_profiled.log 2>&1
python $R/tools/prof_summary.py $(ls /tmp/p_shac/*/*kernel_stats.csv | head -1) $O/r04_shac_kernel_stats.txt "python bench.py --workload shac --steps 256" > /dev/null 2>&1
head -30 $O/r04_shac_kernel_stats.txt
    ;;
b14)
mkdir -p gpurun_out/r04b14
timeout 600 python bench.py --workload bptt --steps 128 2>&1 | tail -1 | tee gpurun_out/r04b14/bench_bptt.json
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r04b14/bench_default.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04b14/bench_default.json'))
print(d['value'], d['roofline']['frac'], {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d.get('secondary',{}).items()})
PY
    ;;
b15)
# r04 batch 15: k_mlp_wgrad at HBM-streaming row counts (SHAC critic 524 288 rows, BPTT horizon 1 M rows): one vs two waves per SIMD
mkdir -p gpurun_out/r04b15
for w in 1 2; do
  for wl in shac bptt; do
    VISFLY_AMD_WGRAD_WPS=$w timeout 600 python bench.py --workload $wl --steps 256 2>&1 | tail -1 > gpurun_out/r04b15/${wl}_wps$w.json
    python - <<PY
import json
d=json.load(open('gpurun_out/r04b15/${wl}_wps$w.json'))
print('$wl wps=$w', d['value'], d['s_per_iteration'])
PY
  done
done
    ;;
b16)
mkdir -p gpurun_out/r04b16
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r04b16/pytest.txt
for wl in shac bptt ppo; do
  timeout 600 python bench.py --workload $wl --steps 256 2>&1 | tail -1 > gpurun_out/r04b16/$wl.json
  python - <<PY
import json
d=json.load(open('gpurun_out/r04b16/$wl.json'))
print('$wl', d['value'], d['s_per_iteration'], d['roofline']['frac'])
PY
done
    ;;
b17)
# state of HEAD after the four-lanes-per-agent BPTT launches: full GPU suite, the driver's bench command, BPTT phases, trainer legs
O=$GRAFT_REPO_ROOT/gpurun_out/r04b17; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; tail -3 $O/pytest_all.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu > $O/bench_default.txt
timeout 300 python tools/exp_bptt_phases.py > $O/bptt_phases.txt 2>&1; cat $O/bptt_phases.txt
for wl in bptt shac ppo; do
  steps=256; [ $wl = bptt ] && steps=128
  timeout 600 python bench.py --workload $wl --steps $steps 2>&1 | tail -1 > $O/$wl.json
done
python - <<'PY'
import json,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04b17/'
for l in open(O+'bench_default.txt'):
    if l.startswith('{'):
        j=json.loads(l); print('default value %.4e'%j['value'], 'ms', j['ms_per_step'], 'frac', round(j['roofline']['frac'],3), 'kernel_us', j['roofline'].get('kernel_us'))
        for k,v in j.get('secondary',{}).items(): print('   ', k, '%.4e'%v['value'], round(v['roofline']['frac'],3))
for wl in ('bptt','shac','ppo'):
    try:
        d=json.load(open(O+wl+'.json')); print(wl, '%.4e'%d['value'], d.get('s_per_iteration'), round(d['roofline']['frac'],3))
    except Exception as e: print(wl, 'failed', e)
PY
    ;;
b18)
O=$GRAFT_REPO_ROOT/gpurun_out/r04b18; mkdir -p $O; cd $GRAFT_REPO_ROOT
for dt in 0.0025 0.005 0.01 0.02; do for q in 0 1; do VF_EXP_DT=$dt VISFLY_AMD_ENV_QUAD=$q timeout 300 python tools/exp_env_quad.py 65536 2>&1 | grep QUAD | tee -a $O/quad_dt.txt; done; done
    ;;
b19)
O=$GRAFT_REPO_ROOT/gpurun_out/r04b19; mkdir -p $O; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
VF_ALT_LIB=$PWD/tools/tmp/libvf_nopf.so timeout 300 python tools/exp_env_quad.py 65536 1048576 2>&1 | grep QUAD | sed 's/^/nopf  /' | tee -a $O/ab.txt
timeout 300 python tools/exp_env_quad.py 65536 1048576 2>&1 | grep QUAD | sed 's/^/pf    /' | tee -a $O/ab.txt
done
VF_ALT_LIB=$PWD/tools/tmp/libvf_trace.so timeout 300 python tools/exp_env_timeline.py 2>&1 | grep -v amdgpu | grep -v "^launch" | tee $O/timeline_pf.txt
timeout 900 python -m pytest tests/test_env_gpu.py tests/test_dyn_gpu.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
    ;;
b20)
# k_env_step / k_dyn_step with preloaded kernel arguments + one-batch constant prefetch: full GPU suite, the driver's bench command, A/B, timeline
O=$GRAFT_REPO_ROOT/gpurun_out/r04b20; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; grep -E "passed|failed|error" $O/pytest_all.txt | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu > $O/bench_default.txt
python - <<'PY'
import json,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04b20/'
for l in open(O+'bench_default.txt'):
    if l.startswith('{'):
        j=json.loads(l); print('default value %.4e'%j['value'], 'ms', j['ms_per_step'], 'frac', round(j['roofline']['frac'],3), 'kernel_us', j['roofline'].get('kernel_us'), 'dyn_only', j['roofline']['dyn_only']['kernel_us'], 'sustained %.3e'%j['sustained']['value'], 'resets %.3e x%.3f'%(j['with_resets']['value'], j['with_resets']['kernel_us_vs_headline']))
        for k,v in j.get('secondary',{}).items(): print('   ', k, '%.4e'%v['value'], round(v['roofline']['frac'],3))
PY
VF_ALT_LIB=$PWD/tools/tmp/libvf_nopf.so timeout 300 python tools/exp_env_quad.py 65536 2>&1 | grep QUAD | sed 's/^/before  /' | tee -a $O/ab.txt
timeout 300 python tools/exp_env_quad.py 65536 2>&1 | grep QUAD | sed 's/^/after   /' | tee -a $O/ab.txt
VF_ALT_LIB=$PWD/tools/tmp/libvf_trace.so timeout 300 python tools/exp_env_timeline.py > $O/timeline.txt 2>&1
    ;;
b21)
# A/B: the granules the interval finalises + the observation row stored BEFORE the epilogue's arithmetic, plain / sc1 / nt (-DVF_EXP_EARLY=1/2/3)
O=$GRAFT_REPO_ROOT/gpurun_out/r04b21; mkdir -p $O; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
timeout 300 python tools/exp_env_quad.py 65536 1048576 2>&1 | grep QUAD | sed 's/^/default /' | tee -a $O/ab.txt
for m in 1 2 3; do VF_ALT_LIB=$PWD/tools/tmp/libvf_early$m.so timeout 300 python tools/exp_env_quad.py 65536 1048576 2>&1 | grep QUAD | sed "s/^/early$m  /" | tee -a $O/ab.txt; done
done
for m in 1 2; do VF_ALT_LIB=$PWD/tools/tmp/libvf_early$m.so timeout 600 python -m pytest tests/test_env_gpu.py -x -q 2>&1 | tail -2 | tee -a $O/pytest.txt; done
    ;;
b22)
# launch time per BASELINE configuration before / after the preloaded-argument treatment of the two-wave split kernel; env tests
O=$GRAFT_REPO_ROOT/gpurun_out/r04b22; mkdir -p $O; cd $GRAFT_REPO_ROOT
VF_ALT_LIB=$PWD/tools/tmp/libvf_nopf.so timeout 600 python tools/exp_configs.py 2>&1 | grep "N=" | sed 's/^/r04a  /' | tee $O/configs.txt
VF_ALT_LIB=$PWD/tools/tmp/libvf_before_split.so timeout 600 python tools/exp_configs.py 2>&1 | grep "N=" | sed 's/^/before /' | tee -a $O/configs.txt
timeout 600 python tools/exp_configs.py 2>&1 | grep "N=" | sed 's/^/after  /' | tee -a $O/configs.txt
timeout 1200 python -m pytest tests/test_env_gpu.py tests/test_dyn_gpu.py tests/test_env_multistep_gpu.py tests/test_config_scale_gpu.py tests/test_env_external_scene_gpu.py -x -q 2>&1 | tail -2 | tee $O/pytest.txt
    ;;
b23)
O=$GRAFT_REPO_ROOT/gpurun_out/r04b23; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 600 python tools/exp_bptt_overlap.py 2>&1 | grep -v amdgpu | tail -5 | tee $O/overlap.txt
timeout 600 python -m pytest tests/test_dyn_gpu.py -x -q 2>&1 | tail -2
    ;;
b24)
O=$GRAFT_REPO_ROOT/gpurun_out/r04b24; mkdir -p $O; cd $GRAFT_REPO_ROOT
for w in 0 1000; do
  ROC_ACTIVE_WAIT_TIMEOUT=$w timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>&1 | grep "^{" > $O/bench_wait$w.json
  python - <<PY
import json
j=json.load(open('$O/bench_wait$w.json')); t=j['timing']
print('ROC_ACTIVE_WAIT_TIMEOUT=$w value %.4e'%j['value'], 'ms/step', round(j['ms_per_step']*1e3,3), 'at completion', round(t['ms_per_step_at_completion']*1e3,3), 'kernel', round(j['roofline']['kernel_us'],3), 'resets %.4e'%j['with_resets']['value'], 'sustained %.4e'%j['sustained']['value'])
PY
done
    ;;
b25)
O=$GRAFT_REPO_ROOT/gpurun_out/r04b25; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ppo_gpu.py -x -q 2>&1 | tail -2 | tee $O/pytest.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/p_ppo -- python $GRAFT_REPO_ROOT/bench.py --workload ppo --steps 256 > $O/bench_ppo.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $(ls /tmp/p_ppo/*/*kernel_stats.csv | head -1) $O/ppo_kernel_stats.txt "python bench.py --workload ppo --steps 256" | head -14
tail -1 $O/bench_ppo.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ppo %.4e'%d['value'], d['s_per_iteration'], d['roofline']['frac'])"
    ;;
b26)
# SHAC: a critic update's forward + loss + reverse chain as one launch (vf_twin_q_update): tests, A/B of the iteration time
O=$GRAFT_REPO_ROOT/gpurun_out/r04b26; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_shac_gpu.py tests/test_abi.py -x -q 2>&1 | tail -8 | tee $O/pytest.txt
for f in 0 1 0 1; do
  VISFLY_AMD_FUSED_CRITIC=$f timeout 600 python bench.py --workload shac --steps 256 2>&1 | tail -1 > $O/shac_$f.json
  python -c "
import json; d=json.load(open('$O/shac_$f.json')); print('fused_critic=$f  %.4e env-steps/s  %.3f ms per iteration  frac %.3f' % (d['value'], d['s_per_iteration']*1e3, d['roofline']['frac']))" | tee -a $O/ab.txt
done
    ;;
b27)
O=$GRAFT_REPO_ROOT/gpurun_out/r04b27; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/p_shac -- python $GRAFT_REPO_ROOT/bench.py --workload shac --steps 256 > $O/bench_shac.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $(ls /tmp/p_shac/*/*kernel_stats.csv | head -1) $O/r04_shac_kernel_stats.txt "python bench.py --workload shac --steps 256" | head -16
cd $GRAFT_REPO_ROOT; timeout 600 python -m pytest tests/test_shac_gpu.py -x -q -rs 2>&1 | tail -4
    ;;
b28)
# PPO: the next epoch's shuffle (randperm + row gather) on a side stream beside the current epoch's optimiser steps: tests + A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r04b28; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_ppo_gpu.py tests/test_parallel_gpu.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
for f in 0 1 0 1; do
  VISFLY_AMD_OVERLAP_SHUFFLE=$f timeout 600 python bench.py --workload ppo --steps 256 2>&1 | tail -1 > $O/ppo_$f.json
  python -c "
import json; d=json.load(open('$O/ppo_$f.json')); print('overlap_shuffle=$f  %.4e env-steps/s  %.2f ms per iteration  frac %.3f' % (d['value'], d['s_per_iteration']*1e3, d['roofline']['frac']), d.get('split_ms'))" | tee -a $O/ab.txt
done
    ;;
b29)
# SHAC: the horizon's 32 target-critic passes + 32 accumulate launches as one launch each (vf_mlp_forward_steps, vf_shac_accumulate_horizon): tests, A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r04b29; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_shac_gpu.py -x -q 2>&1 | tail -4 | tee $O/pytest.txt
for f in 0 1 0 1; do
  VISFLY_AMD_BATCH_TARGETS=$f timeout 600 python bench.py --workload shac --steps 256 2>&1 | tail -1 > $O/shac_$f.json
  python -c "
import json; d=json.load(open('$O/shac_$f.json')); print('batch_targets=$f  %.4e env-steps/s  %.3f ms per iteration  frac %.3f' % (d['value'], d['s_per_iteration']*1e3, d['roofline']['frac']))" | tee -a $O/ab.txt
done
    ;;
b30)
# k_mlp_wgrad in the streaming regime: waves per layer by bytes per row instead of by MFMA tiles -- SHAC / BPTT A/B + tests
O=$GRAFT_REPO_ROOT/gpurun_out/r04b30; mkdir -p $O; cd $GRAFT_REPO_ROOT
for b in tiles bytes tiles bytes; do
  for w in shac bptt; do
  steps=256; [ $w = bptt ] && steps=128
  VISFLY_AMD_WGRAD_BALANCE=$b timeout 600 python bench.py --workload $w --steps $steps 2>&1 | tail -1 > $O/${w}_$b.json
  python -c "
import json; d=json.load(open('$O/${w}_$b.json')); print('$w balance=$b  %.4e env-steps/s  %.3f ms per iteration  frac %.3f' % (d['value'], d['s_per_iteration']*1e3, d['roofline']['frac']))" | tee -a $O/ab.txt
  done
done
timeout 900 python -m pytest tests/test_shac_gpu.py tests/test_bptt_gpu.py tests/test_ppo_gpu.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
    ;;
b31)
# k_wgrad_fold with 16 instead of 4 waves per block (the partial rows split 16 ways): PPO A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r04b31; mkdir -p $O; cd $GRAFT_REPO_ROOT
for v in 4 16 4 16; do
  L=""; [ $v = 16 ] && L=$PWD/tools/tmp/libvf_fold16.so
  VF_ALT_LIB=$L timeout 600 python tools/bench_alt.py --workload ppo --steps 256 2>&1 | tail -1 > $O/ppo_$v.json
  python -c "
import json; d=json.load(open('$O/ppo_$v.json')); print('fold waves=$v  %.4e env-steps/s  %.2f ms per iteration  frac %.3f' % (d['value'], d['s_per_iteration']*1e3, d['roofline']['frac']))" | tee -a $O/ab.txt
done
    ;;
b32)
# k_twin_q_update_chain keeping the ReLU masks as bits (16 per tile) and capped at 256 VGPRs (two waves per SIMD) vs uncapped (370 VGPRs, one wave): SHAC A/B + tests
O=$GRAFT_REPO_ROOT/gpurun_out/r04b32; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_shac_gpu.py -x -q 2>&1 | grep -E "passed|failed|error" | tee $O/pytest.txt
for v in 1 2 1 2; do
  L=""; [ $v = 1 ] && L=$PWD/tools/tmp/libvf_twinq1.so
  VF_ALT_LIB=$L timeout 600 python tools/bench_alt.py --workload shac --steps 256 2>&1 | tail -1 > $O/shac_$v.json
  python -c "
import json; d=json.load(open('$O/shac_$v.json')); print('twin-q waves per SIMD=$v  %.4e env-steps/s  %.3f ms per iteration  frac %.3f' % (d['value'], d['s_per_iteration']*1e3, d['roofline']['frac']))" | tee -a $O/ab.txt
done
    ;;
b33)
O=$GRAFT_REPO_ROOT/gpurun_out/r04b33; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pmm_shac -- python $R/bench.py --workload shac --steps 256 > $O/log_mfma_shac.txt 2>&1
for k in k_twin_q_update_chain k_mlp_wgrad k_mlp_forward_chain k_bptt_rollout k_bptt_reverse; do echo "== $k (shac)" >> $O/pmc_mfma_shac.txt; python $R/tools/pmc_summary.py /tmp/pmm_shac $k >> $O/pmc_mfma_shac.txt 2>&1; done
timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d /tmp/pmf_shac -- python $R/bench.py --workload shac --steps 256 > $O/log_f_shac.txt 2>&1
timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d /tmp/pmw_shac -- python $R/bench.py --workload shac --steps 256 > $O/log_w_shac.txt 2>&1
for k in k_twin_q_update_chain k_mlp_wgrad; do echo "== $k traffic" >> $O/pmc_mfma_shac.txt; python $R/tools/pmc_summary.py /tmp/pmf_shac $k >> $O/pmc_mfma_shac.txt; python $R/tools/pmc_summary.py /tmp/pmw_shac $k >> $O/pmc_mfma_shac.txt; done
cat $O/pmc_mfma_shac.txt
    ;;
b34)
# A/B of the unlikely-branch layout of k_env_step (helper test, episode-end blocks laid out behind the hot path) in both regimes, bench's own method too
O=$GRAFT_REPO_ROOT/gpurun_out/r04b34; mkdir -p $O; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
VF_ALT_LIB=$PWD/tools/tmp/libvf_nounlikely.so timeout 300 python tools/exp_env_quad.py 65536 2>&1 | grep QUAD | sed 's/^/inline   /' | tee -a $O/ab.txt
timeout 300 python tools/exp_env_quad.py 65536 2>&1 | grep QUAD | sed 's/^/unlikely /' | tee -a $O/ab.txt
done
for v in inline unlikely inline unlikely; do
  L=""; [ $v = inline ] && L=$PWD/tools/tmp/libvf_nounlikely.so
  VF_ALT_LIB=$L timeout 600 python tools/bench_alt.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>&1 | grep "^{" > $O/bench_$v.json
  python -c "
import json; j=json.load(open('$O/bench_$v.json')); print('$v  value %.4e kernel %.2f | resets %.4e kernel %.2f x%.3f' % (j['value'], j['roofline']['kernel_us'], j['with_resets']['value'], j['with_resets']['kernel_us'], j['with_resets']['kernel_us_vs_headline']))" | tee -a $O/ab.txt
done
    ;;
b35)
# where does the two-wave split kernel cross the one-lane kernel now?  (both got the preloaded arguments)
O=$GRAFT_REPO_ROOT/gpurun_out/r04b35; mkdir -p $O; cd $GRAFT_REPO_ROOT
for s in 0 1; do VISFLY_AMD_SPLIT=$s timeout 300 python tools/exp_env_quad.py 16384 32768 40960 49152 65536 131072 2>&1 | grep QUAD | sed "s/^/split=$s /" | tee -a $O/split.txt; done
    ;;
b36)
# vf_env.hip / vf_dyn.hip compiled with -mllvm -slp-threshold=8 (sub-step loop 315 -> 301 instructions: fewer packed pairs whose operands need moves): A/B + bit-exactness
O=$GRAFT_REPO_ROOT/gpurun_out/r04b36; mkdir -p $O; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
timeout 300 python tools/exp_env_quad.py 65536 2>&1 | grep QUAD | sed 's/^/default /' | tee -a $O/ab.txt
VF_ALT_LIB=$PWD/tools/tmp/libvf_slp8.so timeout 300 python tools/exp_env_quad.py 65536 2>&1 | grep QUAD | sed 's/^/slp8    /' | tee -a $O/ab.txt
done
VF_ALT_LIB=$PWD/tools/tmp/libvf_slp8.so timeout 300 python tools/exp_configs.py 2>&1 | grep "N=" | sed 's/^/slp8    /' | tee -a $O/ab.txt
timeout 300 python tools/exp_configs.py 2>&1 | grep "N=" | sed 's/^/default /' | tee -a $O/ab.txt
    ;;
b37)
# final env-step evidence on the shipped build (preloaded arguments + SLP threshold 8): rocprofv3 kernel stats, traffic and SQ counters
O=$GRAFT_REPO_ROOT/gpurun_out/r04b37; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/p_env -- python $R/bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --no-reset-leg --sustain-s 1 > $O/bench_env_profiled.log 2>&1
python $R/tools/prof_summary.py $(ls /tmp/p_env/*/*kernel_stats.csv | head -1) $O/r04_env_step_kernel_stats.txt "python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --no-reset-leg --sustain-s 1" | head -6
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d /tmp/pm_f -- python $R/tools/exp_env_one.py 65536 12 > $O/log_f.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d /tmp/pm_w -- python $R/tools/exp_env_one.py 65536 12 > $O/log_w.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pm_a -- python $R/tools/exp_env_one.py 65536 12 > $O/log_a.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -d /tmp/pm_b -- python $R/tools/exp_env_one.py 65536 12 > $O/log_b.txt 2>&1
for p in f w a b; do python $R/tools/pmc_summary.py /tmp/pm_$p k_env_step >> $O/pmc_env.txt 2>&1; done; cat $O/pmc_env.txt
    ;;
final)
# r04 final evidence: full GPU suite, the driver's bench command, rocprofv3 kernel stats of the four workloads, PMC passes (env SQ + MFMA)
O=$GRAFT_REPO_ROOT/gpurun_out/r04f; mkdir -p $O; rm -f $O/pmc_*.txt; R=$GRAFT_REPO_ROOT; cd $R
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
    ;;
final2)
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
    ;;
profiles)
# r04 evidence for profiles/: rocprofv3 kernel-trace stats of the four bench workloads (the commands the bench line's figures come from)
O=$GRAFT_REPO_ROOT/gpurun_out/r04f; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
RP="rocprofv3 --output-format csv"
timeout 600 $RP --kernel-trace --stats -d /tmp/p_env -- python $R/bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --no-reset-leg --sustain-s 1 > $O/bench_env_profiled.log 2>&1
python $R/tools/prof_summary.py $(ls /tmp/p_env/*/*kernel_stats.csv | head -1) $O/r04_env_step_kernel_stats.txt "python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --no-reset-leg --sustain-s 1" > /dev/null 2>&1
for w in ppo bptt shac; do
  steps=256; [ $w = bptt ] && steps=128
  timeout 600 $RP --kernel-trace --stats -d /tmp/p_$w -- python $R/bench.py --workload $w --steps $steps > $O/bench_${w}_profiled.log 2>&1
  python $R/tools/prof_summary.py $(ls /tmp/p_$w/*/*kernel_stats.csv | head -1) $O/r04_${w}_kernel_stats.txt "python bench.py --workload $w --steps $steps" > /dev/null 2>&1
done
head -8 $O/r04_env_step_kernel_stats.txt; head -12 $O/r04_ppo_kernel_stats.txt; head -12 $O/r04_bptt_kernel_stats.txt; head -24 $O/r04_shac_kernel_stats.txt
    ;;
*) echo "usage: $0 b1..b37|final|final2|profiles"; exit 2;;
esac
