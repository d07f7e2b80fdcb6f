#!/bin/bash
# vf_env.hip / vf_dyn.hip compiled with -mllvm -slp-threshold=8 (sub-step loop 315 -> 301 instructions: fewer packed pairs whose operands need moves): A/B + bit-exactness
O=$GRAFT_REPO_ROOT/gpurun_out/r04b36; mkdir -p $O; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
timeout 300 python tools/exp_env_quad.py 65536 2>&1 | grep QUAD | sed 's/^/default /' | tee -a $O/ab.txt
VF_ALT_LIB=$PWD/tools/tmp/libvf_slp8.so timeout 300 python tools/exp_env_quad.py 65536 2>&1 | grep QUAD | sed 's/^/slp8    /' | tee -a $O/ab.txt
done
VF_ALT_LIB=$PWD/tools/tmp/libvf_slp8.so timeout 300 python tools/exp_configs.py 2>&1 | grep "N=" | sed 's/^/slp8    /' | tee -a $O/ab.txt
timeout 300 python tools/exp_configs.py 2>&1 | grep "N=" | sed 's/^/default /' | tee -a $O/ab.txt
