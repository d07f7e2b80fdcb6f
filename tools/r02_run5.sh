#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r02e
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for split in 0 1; do
  VISFLY_AMD_SPLIT=$split timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pmc_a_$split -- python $GRAFT_REPO_ROOT/tools/exp_env_one.py 65536 12 > $O/log_a_$split.txt 2>&1
  VISFLY_AMD_SPLIT=$split timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -d /tmp/pmc_b_$split -- python $GRAFT_REPO_ROOT/tools/exp_env_one.py 65536 12 > $O/log_b_$split.txt 2>&1
  find /tmp/pmc_a_$split /tmp/pmc_b_$split -type f | head -20 >> $O/files.txt
  for p in a b; do python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_${p}_$split k_env_step >> $O/pmc_split$split.txt 2>&1; done
done
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_20.log 2>&1
cat $O/files.txt | head; cat $O/pmc_split0.txt $O/pmc_split1.txt; tail -3 $O/log_a_0.txt
