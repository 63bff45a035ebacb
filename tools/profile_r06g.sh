#!/bin/bash
# Round 6: dense kernel, two workgroups of four waves per CU with two-tile chunks (development switches AOC_DENSE_WAVES=4 AOC_DENSE_NB=2) against the
# product (8 waves, four-tile chunks) and the lone-wave form (4 waves, four-tile chunks, one workgroup per CU).  Output: gpurun_out/r06b/dense_4x2.txt
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
export POOL_STRIDE=5 QUERY_OFFSET=3
{
for rep in 1 2; do
for v in "8 4" "4 4" "4 2"; do
  set -- $v
  for R in 2 6 12; do
  echo "== AOC_DENSE_WAVES=$1 AOC_DENSE_NB=$2 R=$R"
  AOC_DENSE_WAVES=$1 AOC_DENSE_NB=$2 AOC_LIB_VARIANT=dev python tools/bench_dense.py $R 2>&1 | grep -E "^split|max|rescored"
  done
done
done
} > "$out/dense_4x2.txt" 2>&1
cat "$out/dense_4x2.txt"
