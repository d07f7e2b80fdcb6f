#!/bin/bash
# final r02 evidence: profiles (tools/r02_profiles.sh), the reset-regime and chain probes, the full -m gpu suite, the two bench lines
cd $GRAFT_REPO_ROOT
bash tools/r02_profiles.sh > /dev/null 2>&1
O=$GRAFT_REPO_ROOT/gpurun_out/r02q
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 120 tools/mfma_chain_probe > $O/mfma_chain_probe.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" >> $O/status
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/status
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.log 2>&1
echo "bench rc=$?" >> $O/status
timeout 600 python bench.py > $O/bench_default.log 2>&1
echo "bench_default rc=$?" >> $O/status
grep -E "passed|failed" $O/pytest_all.log | tail -2; cat $O/status
