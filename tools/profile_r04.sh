#!/bin/bash
# Round-4 measurement artifacts (GPU box): default bench line, rocprofv3 kernel statistics of the bench command and of the k-means chain
# alone, FETCH / WRITE counter passes of the chain (separate --pmc runs), chain timings, the sequence-sharded evaluation line.
# Also: the pieces of a frame on its stream (bench --segments), the gates and the host cost of the enqueue calls alone, FETCH / WRITE of the gates.
# Usage: tools/profile_r04.sh   (outputs under gpurun_out/r04/)
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
rm -f "$out/kmeans_chain_standalone.txt" "$out/pmc_kmeans_R6_F3.txt" "$out/pmc_gates_cfg2.txt"
python bench.py > "$out/bench_line.json" 2> "$out/bench_line.err"
python bench.py --segments --no-extras --no-cpu-baseline --exact-steps 0 2> /dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(json.dumps(dict(value=d['value'], ms_per_step=d['ms_per_step'], host_enqueue_ms_per_step=d['host_enqueue_ms_per_step'], frame_segments_ms=d['frame_segments_ms']), indent=1))" > "$out/bench_segments.json"
python tools/bench_gates.py > "$out/gates_standalone.txt" 2> /dev/null
python tools/host_cost.py 2> /dev/null | grep "FrameRunner" > "$out/host_cost.txt"
python bench.py --steps 20 --no-extras --no-cpu-baseline > "$out/bench_line_steps20.json" 2> /dev/null
python bench.py --eval-sharded --no-cpu-baseline > "$out/eval_sharded_line.json" 2> /dev/null
for R in 1 6 12; do python tools/bench_kmeans.py $R 1 2>&1 | grep "k-means chain" >> "$out/kmeans_chain_standalone.txt"; done
python tools/bench_kmeans.py 6 3 2>&1 | grep "k-means chain" >> "$out/kmeans_chain_standalone.txt"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --exact-steps 0 > "$out/bench_under_rocprof.json" 2> /dev/null
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_default_bench.csv"; fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_km -- python $GRAFT_REPO_ROOT/tools/bench_kmeans.py 6 3 > /dev/null 2>&1
f=$(find /tmp/prof_km -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_kmeans_chain_R6_F3.csv"; fi
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_km
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_km -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_kmeans.py 6 3 > /tmp/pmc_km.log 2>&1
  python3 - "$ctr" >> "$out/pmc_kmeans_R6_F3.txt" <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/pmc_km/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); raise SystemExit
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"]
    if "km_" in n and r["Counter_Name"] == sys.argv[1]:
        acc[n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{sys.argv[1]:10s} {k:46s} n={len(v):4d} avg={sum(v)/len(v):10.1f} KB")
PY
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_g
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_g -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_gates.py --reps 6 > /tmp/pmc_g.log 2>&1
  python3 - "$ctr" >> "$out/pmc_gates_cfg2.txt" <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/pmc_g/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); raise SystemExit
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"]
    if any(k in n for k in ("film_", "cond_", "plane_mean", "head_delta")) and r["Counter_Name"] == sys.argv[1]:
        key = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40] + " grid=" + r.get("Grid_Size", "?")
        acc[key].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{sys.argv[1]:10s} {k:58s} n={len(v):4d} avg={sum(v)/len(v):10.1f} KB")
PY
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_eval -- python $GRAFT_REPO_ROOT/bench.py --eval-sharded --no-cpu-baseline --eval-scale 0.015 > /dev/null 2>&1
f=$(find /tmp/prof_eval -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_eval_sharded.csv"; fi
ls -la "$out"
