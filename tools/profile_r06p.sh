#!/bin/bash
# Round 6: the bench with and without the bound seeds (libaoc_hip.so = seeds on; libaoc_hip_noseed.so = -DAOC_DENSE_SEED=0), alternating runs on one box.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
{
for rep in 1 2 3; do
for lib in libaoc_hip.so libaoc_hip_q4v.so; do
  for cfg in cfg2 cfg3 cfg4; do
  echo "== bench $cfg $lib"
  AOC_LIB_FILE=$lib python bench.py --config $cfg --no-extras --no-cpu-baseline --exact-steps 0 --details-file gpurun_out/r06b/bd_tmp.json 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], 'frames/s', d['roofline']['avg_launch_ms'], 'ms dense in-run')"
  done
done
done
} > "$out/bench_seed.txt" 2>&1
cat "$out/bench_seed.txt"
