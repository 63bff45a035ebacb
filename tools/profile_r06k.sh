#!/bin/bash
# Round 6: what does dense_prune_q4_kernel spend per tile?  Variant libraries with parts compiled out (WRONG results): 2 no rescoring, 6 + no row DMA, 14 + no barrier.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b; mkdir -p "$out"
cd $GRAFT_REPO_ROOT
export POOL_STRIDE=5 QUERY_OFFSET=3
{
for lib in libaoc_hip_q4v.so libaoc_hip_q4v_d2.so libaoc_hip_q4v_d6.so libaoc_hip_q4v_d14.so; do
  for R in 2 6 12; do
  echo "== $lib R=$R"
  AOC_LIB_FILE=$lib timeout 120 python tools/bench_dense.py $R 2>&1 | grep -E "^split"
  done
done
for dbg in; do
  for R in 2 6 12; do
  echo "== dev 8x2 AOC_DENSE_DEBUG=$dbg R=$R"
  AOC_LIB_VARIANT=dev AOC_DENSE_DEBUG=$dbg timeout 120 python tools/bench_dense.py $R 2>&1 | grep -E "^split"
  done
done
} > "$out/dense_q4_parts.txt" 2>&1
cat "$out/dense_q4_parts.txt"
