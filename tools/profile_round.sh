#!/bin/bash
# Developer tool (GPU box): default bench line + rocprofv3 kernel stats of the same command + PMC traffic passes.
# Usage: tools/profile_round.sh <tag>     (outputs under gpurun_out/<tag>/)
set -u
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 600 python $GRAFT_REPO_ROOT/bench.py > "$out/bench.json" 2> "$out/bench.err"
tail -c 600 "$out/bench.err"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $GRAFT_REPO_ROOT/bench.py > "$out/bench_under_rocprof.json" 2> /dev/null
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$out/kernel_stats.csv"; fi
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_$ctr -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /dev/null 2>&1
  f=$(find /tmp/prof_$ctr -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" "$ctr" > "$out/pmc_$ctr.txt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") != sys.argv[2]:
        continue
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    acc[k][0] += 1
    acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k:62s} dispatches {n:6d}  sum {v:16.1f}  per-dispatch {v / n:14.2f}")
PY
  fi
done
ls -la "$out"
