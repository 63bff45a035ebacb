#!/bin/bash
# Round 6: dense kernel experiments (development build): 2048 burst priority, 3072 asymmetric burst priorities, 16384 row DMA issued between the tiles.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
export POOL_STRIDE=5 QUERY_OFFSET=3
{
for rep in 1 2; do
for dbg in 0 2048 3072 16384 18432 19456; do
  echo "== AOC_DENSE_DEBUG=$dbg"
  AOC_LIB_VARIANT=dev AOC_DENSE_DEBUG=$dbg python tools/bench_dense.py 6 2>&1 | grep -E "^split|max"
done
done
} > "$out/dense_exp_e.txt" 2>&1
cat "$out/dense_exp_e.txt"
