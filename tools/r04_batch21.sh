#!/bin/bash
# A/B: the granules the interval finalises + the observation row stored BEFORE the epilogue's arithmetic, plain / sc1 / nt (-DVF_EXP_EARLY=1/2/3)
O=$GRAFT_REPO_ROOT/gpurun_out/r04b21; mkdir -p $O; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
timeout 300 python tools/exp_env_quad.py 65536 1048576 2>&1 | grep QUAD | sed 's/^/default /' | tee -a $O/ab.txt
for m in 1 2 3; do VF_ALT_LIB=$PWD/tools/tmp/libvf_early$m.so timeout 300 python tools/exp_env_quad.py 65536 1048576 2>&1 | grep QUAD | sed "s/^/early$m  /" | tee -a $O/ab.txt; done
done
for m in 1 2; do VF_ALT_LIB=$PWD/tools/tmp/libvf_early$m.so timeout 600 python -m pytest tests/test_env_gpu.py -x -q 2>&1 | tail -2 | tee -a $O/pytest.txt; done
