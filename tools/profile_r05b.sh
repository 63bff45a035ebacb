#!/bin/bash
# Round-5, dense checkpoint A/B (GPU box): dense tests, then tools/bench_dense.py (bench-like pools: stride 5, query 3 frames after the newest
# pool frame) for the development build's AOC_DENSE_CKPT = 0 / 3 / 4 at R = 2 / 6 / 12 (cfg2) and R = 3 (cfg4).
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_dense_split.py tests/test_gpu_frame.py tests/test_gpu_corr_batched.py -x -q -m gpu 2>&1 | tail -5 > "$out/dense_ckpt_tests.txt"
rm -f "$out/dense_ckpt_ab.txt"
for spec in "2 cfg2" "6 cfg2" "12 cfg2" "3 cfg4"; do
  for ck in 0 3 4; do
    echo "== AOC_DENSE_CKPT=$ck R/cfg = $spec" >> "$out/dense_ckpt_ab.txt"
    AOC_LIB_VARIANT=dev AOC_DENSE_CKPT=$ck POOL_STRIDE=5 QUERY_OFFSET=3 python tools/bench_dense.py $spec 2>&1 | grep -v "^fp32" >> "$out/dense_ckpt_ab.txt"
  done
done
