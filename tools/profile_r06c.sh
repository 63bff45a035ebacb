#!/bin/bash
# Round 6: is the dense kernel's MFMA stream paced by the data (power) or by the code?  Development build, pure-stream mode (AOC_DENSE_DEBUG=814: no
# rescoring, no DMA, no barrier, no decisions, no fragment reads) on the real embeddings vs all-zero operands.  Output: gpurun_out/r06b/dense_data.txt
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
export POOL_STRIDE=5 QUERY_OFFSET=3
{
for dbg in 814 2862 46 2094; do
 for scale in 1 0; do
  echo "== AOC_DENSE_DEBUG=$dbg DATA_SCALE=$scale"
  DATA_SCALE=$scale AOC_LIB_VARIANT=dev AOC_DENSE_DEBUG=$dbg python tools/bench_dense.py 6 2>&1 | grep -E "^split"
 done
done
} > "$out/dense_data.txt" 2>&1
cat "$out/dense_data.txt"
