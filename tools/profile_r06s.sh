#!/bin/bash
# Round 6: K = 64 replica-fused assignment (development switch AOC_KM_REP64: a whole CU's LDS for one workgroup, up to four code books) against one replica per item.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
{
for v in 0 1; do
  echo "== AOC_KM_REP64=$v test"
  AOC_KM_REP64=$v AOC_LIB_VARIANT=dev python -m pytest tests/test_gpu_dense_split.py -q -k "replica_fused" 2>&1 | tail -2
  echo "== AOC_KM_REP64=$v chain cfg4 R=3 F=3"
  AOC_KM_REP64=$v AOC_LIB_VARIANT=dev python tools/bench_kmeans_ev.py 3 3 3 10 cfg4 2>/dev/null | tail -1
  AOC_KM_REP64=$v AOC_LIB_VARIANT=dev python tools/bench_kmeans_ev.py 1 2 3 10 cfg4 2>/dev/null | tail -1
done
for rep in 1 2 3; do for v in 0 1; do
  echo "== bench cfg4 AOC_KM_REP64=$v"
  AOC_KM_REP64=$v AOC_LIB_VARIANT=dev python bench.py --config cfg4 --no-extras --no-cpu-baseline --exact-steps 0 --details-file gpurun_out/r06b/bd_tmp.json 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], 'frames/s', d['roofline_kmeans_chain']['avg_launch_ms'], 'ms chain in-run')"
done; done
} > "$out/km_rep64.txt" 2>&1
cat "$out/km_rep64.txt"
