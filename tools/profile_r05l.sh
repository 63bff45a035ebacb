#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
rm -f "$out/dense_lean.txt"
AOC_LIB_VARIANT=dev AOC_DENSE_LEAN=1 python -m pytest tests/test_gpu_dense_split.py -x -q -m gpu 2>&1 | tail -2 >> "$out/dense_lean.txt"
for lean in 0 1; do
  echo "== dense alone AOC_DENSE_LEAN=$lean" >> "$out/dense_lean.txt"
  AOC_LIB_VARIANT=dev AOC_DENSE_LEAN=$lean POOL_STRIDE=5 QUERY_OFFSET=3 python tools/bench_dense.py 6 cfg2 2>&1 | grep "^split" >> "$out/dense_lean.txt"
done
line() { python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d.get('kernels', {}); print('$1', d['value'], 'frames/s', d['ms_per_step'], 'ms/step', 'dense', (d.get('roofline_dense') or {}).get('avg_launch_ms'), 'film', (d.get('roofline_film_scale') or {}).get('avg_launch_ms'), 'cond', (d.get('roofline_cond_gate_pool') or {}).get('avg_launch_ms'))"; }
for cfg in cfg2 cfg4 cfg3; do
  for v in "0 -1" "1 -1" "1 4" "0 -1" "1 -1" "1 4"; do
    set -- $v
    fa=""; if [ "$2" != "-1" ]; then fa="AOC_FILM_AHEAD=$2"; fi
    env AOC_LIB_VARIANT=dev AOC_DENSE_LEAN=$1 $fa python bench.py --config $cfg --no-extras --no-cpu-baseline --exact-steps 0 2> /dev/null | line "$cfg LEAN=$1 FILM_AHEAD=$2" >> "$out/dense_lean.txt"
  done
done
