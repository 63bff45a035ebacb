#!/bin/bash
# Round 6: the dense kernel with FOUR waves per workgroup (development switch AOC_DENSE_WAVES=4: one wave per SIMD, nobody to share the matrix pipe with)
# against the product's eight: what is a lone wave's tile period?  Output: gpurun_out/r06b/dense_waves4.txt
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
export POOL_STRIDE=5 QUERY_OFFSET=3
{
for w in 8 4; do
for dbg in 0 2 46 4096 12288 4098 12290; do
  echo "== AOC_DENSE_WAVES=$w AOC_DENSE_DEBUG=$dbg"
  AOC_DENSE_WAVES=$w AOC_LIB_VARIANT=dev AOC_DENSE_DEBUG=$dbg python tools/bench_dense.py 6 2>&1 | grep -E "^split|dev_cycles|max"
done
done
} > "$out/dense_waves4.txt" 2>&1
cat "$out/dense_waves4.txt"
