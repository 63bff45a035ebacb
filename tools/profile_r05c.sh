#!/bin/bash
# Round-5, speculative fold (GPU box): all bit-exact k-means tests, then the chain alone (hipEvent-timed) with the development build's
# AOC_KM_SPEC = 0 / 1 and a sweep of the literal head chunks under the new tail.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_dense_split.py -x -q -m gpu 2>&1 | tail -5 > "$out/spec_fold_tests.txt"
rm -f "$out/spec_fold_ab.txt"
for rf in "6 1" "6 3" "12 1" "12 3"; do
  for sp in 0 1; do
    echo "== AOC_KM_SPEC=$sp" >> "$out/spec_fold_ab.txt"
    AOC_LIB_VARIANT=dev AOC_KM_SPEC=$sp python tools/bench_kmeans_ev.py $rf 3 10 2> /dev/null | tail -1 >> "$out/spec_fold_ab.txt"
  done
done
for hc in 2 4 6 8 12 16 20; do
  echo "== AOC_KM_SPEC=1 AOC_KM_HEAD_CHUNKS=$hc" >> "$out/spec_fold_ab.txt"
  AOC_LIB_VARIANT=dev AOC_KM_HEAD_CHUNKS=$hc python tools/bench_kmeans_ev.py 6 3 3 10 2> /dev/null | tail -1 >> "$out/spec_fold_ab.txt"
  AOC_LIB_VARIANT=dev AOC_KM_HEAD_CHUNKS=$hc python tools/bench_kmeans_ev.py 12 1 3 10 2> /dev/null | tail -1 >> "$out/spec_fold_ab.txt"
done
python tools/bench_kmeans_ev.py 6 3 3 10 cfg3 2> /dev/null | tail -1 >> "$out/spec_fold_ab.txt"
