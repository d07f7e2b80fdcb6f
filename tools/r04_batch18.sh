#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04b18; mkdir -p $O; cd $GRAFT_REPO_ROOT
for dt in 0.0025 0.005 0.01 0.02; do for q in 0 1; do VF_EXP_DT=$dt VISFLY_AMD_ENV_QUAD=$q timeout 300 python tools/exp_env_quad.py 65536 2>&1 | grep QUAD | tee -a $O/quad_dt.txt; done; done
