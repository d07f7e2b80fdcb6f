#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r02d
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ppo_golden.py tests/test_env_gpu.py -m gpu -q > $O/pytest_fix.log 2>&1
echo "pytest_fix rc=$?" >> $O/status
cd /tmp; export TMPDIR=/tmp
for split in 0 1; do
  VISFLY_AMD_SPLIT=$split timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_a_split$split -- python $GRAFT_REPO_ROOT/tools/exp_env_one.py 65536 12 > $O/pmc_a_split$split.log 2>&1
  VISFLY_AMD_SPLIT=$split timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -d $O/pmc_b_split$split -- python $GRAFT_REPO_ROOT/tools/exp_env_one.py 65536 12 > $O/pmc_b_split$split.log 2>&1
  for p in a b; do python $GRAFT_REPO_ROOT/tools/pmc_summary.py $O/pmc_${p}_split$split k_env_step >> $O/pmc_split$split.txt 2>&1; done
  VISFLY_AMD_SPLIT=$split timeout 120 python $GRAFT_REPO_ROOT/tools/exp_stagger.py 65536 single >> $O/times.log 2>&1
done
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.log 2>&1
echo "bench rc=$?" >> $O/status
rm -rf $O/pmc_a_split* $O/pmc_b_split*
cat $O/pmc_split0.txt $O/pmc_split1.txt; grep -v amdgpu $O/times.log; tail -5 $O/pytest_fix.log
