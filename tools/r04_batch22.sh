#!/bin/bash
# launch time per BASELINE configuration before / after the preloaded-argument treatment of the two-wave split kernel; env tests
O=$GRAFT_REPO_ROOT/gpurun_out/r04b22; mkdir -p $O; cd $GRAFT_REPO_ROOT
VF_ALT_LIB=$PWD/tools/tmp/libvf_nopf.so timeout 600 python tools/exp_configs.py 2>&1 | grep "N=" | sed 's/^/r04a  /' | tee $O/configs.txt
VF_ALT_LIB=$PWD/tools/tmp/libvf_before_split.so timeout 600 python tools/exp_configs.py 2>&1 | grep "N=" | sed 's/^/before /' | tee -a $O/configs.txt
timeout 600 python tools/exp_configs.py 2>&1 | grep "N=" | sed 's/^/after  /' | tee -a $O/configs.txt
timeout 1200 python -m pytest tests/test_env_gpu.py tests/test_dyn_gpu.py tests/test_env_multistep_gpu.py tests/test_config_scale_gpu.py tests/test_env_external_scene_gpu.py -x -q 2>&1 | tail -2 | tee $O/pytest.txt
