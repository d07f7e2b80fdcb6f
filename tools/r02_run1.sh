#!/bin/bash
# r02 GPU session 1: new boundary tests, step_n / graph / two-stream timing, VALU cost probe, bench line
mkdir -p gpurun_out/r02a
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_env_multistep_gpu.py -x -q > gpurun_out/r02a/pytest_new.log 2>&1
echo "pytest_new rc=$?" >> gpurun_out/r02a/status
timeout 300 python tools/exp_step_n.py 65536 20 > gpurun_out/r02a/step_n.log 2>&1
echo "step_n rc=$?" >> gpurun_out/r02a/status
VISFLY_AMD_SPLIT=0 timeout 300 python tools/exp_step_n.py 65536 20 > gpurun_out/r02a/step_n_nosplit.log 2>&1
timeout 120 tools/valu_cost_probe > gpurun_out/r02a/valu_probe.log 2>&1
echo "probe rc=$?" >> gpurun_out/r02a/status
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02a/bench_20.log 2>&1
echo "bench rc=$?" >> gpurun_out/r02a/status
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest_all.log 2>&1
echo "pytest_all rc=$?" >> gpurun_out/r02a/status
tail -5 gpurun_out/r02a/*.log
