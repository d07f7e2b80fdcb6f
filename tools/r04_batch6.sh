#!/bin/bash
# r04 GPU batch 6: k_mlp_wgrad prefetch-ring depth A/B (three builds), env-step launch time vs agents, SAC chain tests
O=$GRAFT_REPO_ROOT/gpurun_out/r04b6; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for lib in visfly_amd/csrc/libvisfly_amd.so tools/libvf_wg96.bin tools/libvf_wg128.bin; do echo "== $lib" >> $O/wgrad_depth.txt; timeout 200 python tools/exp_wgrad_mall.py 25600 30 $R/$lib 2>&1 | grep -v amdgpu | sed -n 2,3p >> $O/wgrad_depth.txt; done; done
cat $O/wgrad_depth.txt
timeout 300 python tools/exp_env_scaling.py 2>&1 | grep -v amdgpu | tee $O/env_scaling.txt
timeout 1500 python -m pytest tests/test_ppo_gpu.py tests/test_shac_gpu.py -x -q -m gpu -k "sac or shac" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
