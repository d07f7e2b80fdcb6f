#!/bin/bash
mkdir -p gpurun_out/r02c
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/r02c/pytest_all.log 2>&1
echo "pytest_all rc=$?" >> gpurun_out/r02c/status
for q in 0 8 16 24 32 48; do VISFLY_AMD_STAGGER=$q timeout 120 python tools/exp_stagger.py 65536 single >> gpurun_out/r02c/stagger.log 2>&1; done
timeout 200 python tools/exp_stagger.py 65536 two >> gpurun_out/r02c/stagger.log 2>&1
VISFLY_AMD_SPLIT=0 timeout 200 python tools/exp_stagger.py 65536 two >> gpurun_out/r02c/stagger.log 2>&1
grep -v amdgpu.ids gpurun_out/r02c/stagger.log
tail -n 25 gpurun_out/r02c/pytest_all.log
