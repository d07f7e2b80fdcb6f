#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04b19; mkdir -p $O; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
VF_ALT_LIB=$PWD/tools/tmp/libvf_nopf.so timeout 300 python tools/exp_env_quad.py 65536 1048576 2>&1 | grep QUAD | sed 's/^/nopf  /' | tee -a $O/ab.txt
timeout 300 python tools/exp_env_quad.py 65536 1048576 2>&1 | grep QUAD | sed 's/^/pf    /' | tee -a $O/ab.txt
done
VF_ALT_LIB=$PWD/tools/tmp/libvf_trace.so timeout 300 python tools/exp_env_timeline.py 2>&1 | grep -v amdgpu | grep -v "^launch" | tee $O/timeline_pf.txt
timeout 900 python -m pytest tests/test_env_gpu.py tests/test_dyn_gpu.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
