#!/bin/bash
# Round 6: dense_prune_q4_kernel (one wave per SIMD, four query tiles, decisions behind the next chain's MFMAs, both planes of a reference tile a tile ahead in
# registers; libaoc_hip_q4.so = -DAOC_DENSE_Q4=1, _q4v = + -mllvm -amdgpu-mfma-vgpr-form=1) against the product kernel.  Output: gpurun_out/r06b/dense_q4.txt
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
export POOL_STRIDE=5 QUERY_OFFSET=3
{
for lib in libaoc_hip.so libaoc_hip_q4.so libaoc_hip_q4v.so libaoc_hip_q4v_d2.so; do
  for R in 1 2 4 6 9 12; do
  echo "== $lib R=$R"
  AOC_LIB_FILE=$lib timeout 120 python tools/bench_dense.py $R 2>&1 | grep -E "^split|max|rescored|Error|error"
  done
done
for lib in libaoc_hip_q4.so libaoc_hip_q4v.so; do
echo "== tests with $lib"
AOC_LIB_FILE=$lib timeout 900 python -m pytest tests/test_gpu_dense_split.py tests/test_gpu_frame.py tests/test_gpu_corr_batched.py -q -x 2>&1 | tail -3
done
} > "$out/dense_q4.txt" 2>&1
cat "$out/dense_q4.txt"
