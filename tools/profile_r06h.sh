#!/bin/bash
# Round 6: dense kernel with the next tile's first FOUR A fragments requested mid-tile (libaoc_hip_pf4.so: -DAOC_DENSE_PF4=1) against the release kernel
# (two fragments, requested after the tile's last MFMA).  Alone (tools/bench_dense.py, bench pools) and in the bench.  Output: gpurun_out/r06b/dense_pf4.txt
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
export POOL_STRIDE=5 QUERY_OFFSET=3
{
for rep in 1 2 3; do
for lib in libaoc_hip.so libaoc_hip_pf4.so; do
  for R in 2 6 12; do
  echo "== $lib R=$R"
  AOC_LIB_FILE=$lib python tools/bench_dense.py $R 2>&1 | grep -E "^split|max"
  done
done
done
for rep in 1 2 3; do
for lib in libaoc_hip.so libaoc_hip_pf4.so; do
  echo "== bench $lib"
  AOC_LIB_FILE=$lib python bench.py --no-extras --no-cpu-baseline --exact-steps 0 --details-file gpurun_out/r06b/bd_tmp.json 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], 'frames/s', d['roofline']['avg_launch_ms'], 'ms dense in-run')"
done
done
} > "$out/dense_pf4.txt" 2>&1
cat "$out/dense_pf4.txt"
