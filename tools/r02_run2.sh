#!/bin/bash
# r02 GPU session 2: full gpu suite after the parity / trainer changes, bench line
mkdir -p gpurun_out/r02b
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r02b/pytest_all.log 2>&1
echo "pytest_all rc=$?" >> gpurun_out/r02b/status
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02b/bench_20.log 2>&1
echo "bench rc=$?" >> gpurun_out/r02b/status
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02b/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r02b/status
tail -n 30 gpurun_out/r02b/pytest_all.log
