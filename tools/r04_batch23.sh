#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04b23; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 600 python tools/exp_bptt_overlap.py 2>&1 | grep -v amdgpu | tail -5 | tee $O/overlap.txt
timeout 600 python -m pytest tests/test_dyn_gpu.py -x -q 2>&1 | tail -2
