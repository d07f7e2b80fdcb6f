#!/bin/bash
# r04 GPU batch 1: fused-weight-gradient skeleton, Infinity-Cache test of k_mlp_wgrad, two waves per SIMD in k_mlp_wgrad (A/B)
O=$GRAFT_REPO_ROOT/gpurun_out/r04b1; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 120 tools/wgrad_fused_probe > $O/wgrad_fused_probe.txt 2>&1
for w in 1 2; do VISFLY_AMD_WGRAD_WPS=$w timeout 200 python tools/exp_wgrad_mall.py 2>&1 | grep -v amdgpu > $O/wgrad_mall_wps$w.txt; done
for w in 1 2; do VISFLY_AMD_WGRAD_WPS=$w timeout 300 python bench.py --workload ppo --steps 256 2>&1 | grep -v amdgpu > $O/bench_ppo_wps$w.txt; done
VISFLY_AMD_WGRAD_WPS=2 timeout 900 python -m pytest tests/test_ppo_gpu.py tests/test_bptt_gpu.py tests/test_shac_gpu.py -x -q -m gpu > $O/pytest_wps2.txt 2>&1
tail -3 $O/pytest_wps2.txt; cat $O/wgrad_fused_probe.txt $O/wgrad_mall_wps*.txt; tail -c 1500 $O/bench_ppo_wps1.txt; echo; tail -c 1500 $O/bench_ppo_wps2.txt
