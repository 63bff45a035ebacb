#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -3 > "$out/spec_fold_tests3.txt"
rm -f "$out/spec_fold_ab3.txt"
for spec in "6 3 cfg2" "6 3 cfg3" "2 3 cfg3" "3 3 cfg4"; do
  set -- $spec
  for sp in 0 1 2; do
    echo "== AOC_KM_SPEC=$sp" >> "$out/spec_fold_ab3.txt"
    AOC_LIB_VARIANT=dev AOC_KM_SPEC=$sp python tools/bench_kmeans_ev.py $1 $2 3 10 $3 2> /dev/null | tail -1 >> "$out/spec_fold_ab3.txt"
  done
done
line() { python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['value'], 'frames/s', d['ms_per_step'], 'ms/step', 'host', d.get('host_enqueue_ms_per_step'))"; }
for cfg in cfg3 cfg4 cfg2; do
  for sp in 0 1 2 -1 0 1 2 -1; do
    AOC_LIB_VARIANT=dev AOC_KM_SPEC=$sp python bench.py --config $cfg --no-extras --no-cpu-baseline --exact-steps 0 2> /dev/null | line "$cfg SPEC=$sp" >> "$out/spec_fold_ab3.txt"
  done
done
