#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r02l
mkdir -p $O
cd $GRAFT_REPO_ROOT
for c in 4 16; do
  for st in 2000; do
    timeout 300 python bench.py --steps $st --warmup 5 --chunk $c --no-secondary --no-cpu-baseline > $O/b_${c}_${st}.log 2>&1
    python - <<PY
import json
l=[x for x in open("$O/b_${c}_${st}.log") if x.startswith("{")][-1]
d=json.loads(l)
print("chunk", $c, "steps", $st, "us/step", round(d["ms_per_step"]*1e3,3), "event", round(d["timing"]["event_us_per_step"],3), "host", round(d["timing"]["host_us_per_step"],2), "per_call", round(d["timing"]["per_call"]["ms_per_step"]*1e3,3), "end_rate", d["timing"]["episode_end_rate"], "steps_with_reset", d["timing"]["steps_with_a_reset"])
PY
  done
done
