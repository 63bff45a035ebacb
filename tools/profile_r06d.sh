#!/bin/bash
# Round 6: core-clock stamps inside the dense kernel (development build, AOC_DENSE_DEBUG 4096 [+ 8192]) with parts switched off.
# Output: gpurun_out/r06b/dense_cycles2.txt
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
export POOL_STRIDE=5 QUERY_OFFSET=3
{
for extra in 0 2 6 10 14 34 46; do
 for sel in 4096 12288; do
  dbg=$((extra + sel))
  echo "== AOC_DENSE_DEBUG=$dbg (= $extra + $sel; 2 no rescoring, 4 no row DMA, 8 no barrier, 32 no bound DMA)"
  AOC_LIB_VARIANT=dev AOC_DENSE_DEBUG=$dbg python tools/bench_dense.py 6 2>&1 | grep -E "^split|dev_cycles"
 done
done
} > "$out/dense_cycles2.txt" 2>&1
cat "$out/dense_cycles2.txt"
