#!/bin/bash
# where does the two-wave split kernel cross the one-lane kernel now?  (both got the preloaded arguments)
O=$GRAFT_REPO_ROOT/gpurun_out/r04b35; mkdir -p $O; cd $GRAFT_REPO_ROOT
for s in 0 1; do VISFLY_AMD_SPLIT=$s timeout 300 python tools/exp_env_quad.py 16384 32768 40960 49152 65536 131072 2>&1 | grep QUAD | sed "s/^/split=$s /" | tee -a $O/split.txt; done
