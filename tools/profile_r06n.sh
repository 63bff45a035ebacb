#!/bin/bash
# Round 6: where does the product dense kernel's fixed cost per launch (0.36 ms) come from?  Development switch AOC_DENSE_ROUNDS = rounds of workgroups (1 -> 5
# splits x 51 query blocks = one round, 2 -> 10 splits (product), 4 -> 20 splits).  Output: gpurun_out/r06b/dense_rounds.txt
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
export POOL_STRIDE=5 QUERY_OFFSET=3
{
for rounds in 1 2 4; do
  for R in 1 2 6 12; do
  echo "== AOC_DENSE_ROUNDS=$rounds R=$R"
  AOC_DENSE_ROUNDS=$rounds AOC_LIB_VARIANT=dev timeout 120 python tools/bench_dense.py $R 2>&1 | grep -E "^split|rescored"
  done
done
for rounds in 1 2; do
  for R in 1 6; do
  echo "== q4 AOC_DENSE_ROUNDS=$rounds R=$R"
  AOC_DENSE_Q4=1 AOC_DENSE_ROUNDS=$rounds AOC_LIB_VARIANT=dev timeout 120 python tools/bench_dense.py $R 2>&1 | grep -E "^split|rescored"
  done
done
} > "$out/dense_rounds.txt" 2>&1
cat "$out/dense_rounds.txt"
