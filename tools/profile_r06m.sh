#!/bin/bash
# Round 6: kernel durations of the dense op at R = 1 / 2 (rocprofv3 --kernel-trace --stats), product kernel vs dense_prune_q4_kernel.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export POOL_STRIDE=5 QUERY_OFFSET=3 PYTHONPATH=$GRAFT_REPO_ROOT
{
for lib in libaoc_hip.so libaoc_hip_q4v.so; do
for R in 1 2 6; do
  rm -rf /tmp/prof_d
  AOC_LIB_FILE=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d -- python $GRAFT_REPO_ROOT/tools/bench_dense.py $R > /dev/null 2>&1
  f=$(find /tmp/prof_d -name "*kernel_stats.csv" | head -1)
  echo "== $lib R=$R"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    if any(k in n for k in ("dense_prune", "split_plan", "finalize", "split_rows", "dense_match", "gate")):
        print(f"  {n:46s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}  max {float(r['MaxNs'])/1e3:9.1f}")
PY
done
done
} > "$out/dense_op_kernels.txt" 2>&1
cat "$out/dense_op_kernels.txt"
