#!/bin/bash
# Round-5: the bench with / without the speculative fold (development build switch), head-chunk variants in-bench.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
rm -f "$out/spec_fold_bench.txt"
line() { python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d.get('kernels', {}); print('$1', d['value'], 'frames/s', d['ms_per_step'], 'ms/step', 'kmeans', d.get('roofline_kmeans_chain', {}).get('avg_ms'), 'host', d.get('host_enqueue_ms_per_step'))"; }
for sp in 0 1 0 1; do
  AOC_LIB_VARIANT=dev AOC_KM_SPEC=$sp python bench.py --no-extras --no-cpu-baseline --exact-steps 0 2> /dev/null | line "cfg2 SPEC=$sp" >> "$out/spec_fold_bench.txt"
done
for hc in 16 24 28; do
  AOC_LIB_VARIANT=dev AOC_KM_HEAD_CHUNKS=$hc python bench.py --no-extras --no-cpu-baseline --exact-steps 0 2> /dev/null | line "cfg2 SPEC=1 heads=$hc" >> "$out/spec_fold_bench.txt"
done
for sp in 0 1; do
  AOC_LIB_VARIANT=dev AOC_KM_SPEC=$sp python bench.py --config cfg3 --no-extras --no-cpu-baseline --exact-steps 0 2> /dev/null | line "cfg3 SPEC=$sp" >> "$out/spec_fold_bench.txt"
  AOC_LIB_VARIANT=dev AOC_KM_SPEC=$sp python bench.py --config cfg4 --no-extras --no-cpu-baseline --exact-steps 0 2> /dev/null | line "cfg4 SPEC=$sp" >> "$out/spec_fold_bench.txt"
done
