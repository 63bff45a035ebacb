#!/bin/bash
# Round 6: seeds of the shared bounds (dense_seed_kernel) on / off: development switch AOC_DENSE_SEED.  Output: gpurun_out/r06b/dense_seed.txt
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
export POOL_STRIDE=5 QUERY_OFFSET=3
{
for seed in 0 1; do
  for R in 1 2 4 6 9 12; do
  echo "== AOC_DENSE_SEED=$seed R=$R"
  AOC_DENSE_SEED=$seed AOC_LIB_VARIANT=dev timeout 120 python tools/bench_dense.py $R 2>&1 | grep -E "^split|rescored|max"
  done
done
echo "== tests (release library: seeds on)"
timeout 900 python -m pytest tests/test_gpu_dense_split.py tests/test_gpu_frame.py tests/test_gpu_corr_batched.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3
} > "$out/dense_seed.txt" 2>&1
cat "$out/dense_seed.txt"
