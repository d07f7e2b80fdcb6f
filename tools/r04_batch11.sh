#!/bin/bash
# r04 batch 11: persistent BPTT launches for the reference actor (td_policies.Actor) -- parity tests + timing
mkdir -p gpurun_out/r04b11
timeout 1200 python -m pytest tests/test_bptt_gpu.py tests/test_shac_gpu.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r04b11/pytest.txt
timeout 600 python tools/exp_bptt_refactor.py 2>&1 | tee gpurun_out/r04b11/ref_actor.txt
