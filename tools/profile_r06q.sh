#!/bin/bash
# Round 6: rounds of dense workgroups (development switch AOC_DENSE_ROUNDS) re-swept with the bound seeds on: alone and in the bench.  Output: gpurun_out/r06b/dense_rounds_seeded.txt
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
{
export POOL_STRIDE=5 QUERY_OFFSET=3
for rounds in 1 2 3 4; do
  for R in 2 6 12; do
  echo "== alone AOC_DENSE_ROUNDS=$rounds R=$R"
  AOC_DENSE_ROUNDS=$rounds AOC_LIB_VARIANT=dev timeout 120 python tools/bench_dense.py $R 2>&1 | grep -E "^split"
  done
done
unset POOL_STRIDE QUERY_OFFSET
for rep in 1 2; do
for rounds in 1 2 3 4; do
  echo "== bench cfg2 AOC_DENSE_ROUNDS=$rounds"
  AOC_DENSE_ROUNDS=$rounds AOC_LIB_VARIANT=dev python bench.py --no-extras --no-cpu-baseline --exact-steps 0 --details-file gpurun_out/r06b/bd_tmp.json 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], 'frames/s', d['roofline']['avg_launch_ms'], 'ms dense in-run')"
done
done
} > "$out/dense_rounds_seeded.txt" 2>&1
cat "$out/dense_rounds_seeded.txt"
