#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r02h
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_env_multistep_gpu.py tests/test_env_gpu.py tests/test_dyn_gpu.py tests/test_config_scale_gpu.py tests/test_bptt_gpu.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/status
timeout 300 python tools/exp_fused.py > $O/fused.log 2>&1
timeout 120 python tools/exp_stagger.py 65536 single >> $O/fused.log 2>&1
grep -v amdgpu $O/fused.log; tail -4 $O/pytest.log
