#!/bin/bash
# r04 batch 12: SHAC's horizon on the persistent launches -- parity + bench leg
mkdir -p gpurun_out/r04b12
timeout 1200 python -m pytest tests/test_bptt_gpu.py tests/test_shac_gpu.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r04b12/pytest.txt
timeout 600 python bench.py --workload shac --steps 256 2>&1 | tail -1 | tee gpurun_out/r04b12/bench_shac.json
