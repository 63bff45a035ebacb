#!/bin/bash
# Round-6 measurement artifacts (GPU box), final code: driver-style bench line (+ --steps 20, + segments), rocprofv3 kernel statistics of the cfg2 /
# cfg3 / cfg4 bench and of the k-means chain alone, FETCH / WRITE counter passes of the chain (separate --pmc runs), hipEvent-timed chains, gates
# alone, the sequence-sharded evaluation line, the step ablations.
# Usage: tools/profile_r05.sh [part ...]   (parts: bench stats pmc chain gates eval ablate; default all; outputs under gpurun_out/r06f/)
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06f
mkdir -p "$out"
parts="${*:-bench stats pmc chain gates eval ablate}"
has() { case " $parts " in *" $1 "*) return 0;; *) return 1;; esac; }
cd $GRAFT_REPO_ROOT
if has bench; then
  python bench.py > "$out/bench_line.json" 2> "$out/bench_line.err"
  python bench.py --steps 20 --no-extras --no-cpu-baseline > "$out/bench_line_steps20.json" 2> /dev/null
  python bench.py --segments --no-extras --no-cpu-baseline --exact-steps 0 2> /dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(json.dumps(dict(value=d['value'], ms_per_step=d['ms_per_step'], host_enqueue_ms_per_step=d['host_enqueue_ms_per_step'], host_enqueue_wall_ms_per_step=d.get('host_enqueue_wall_ms_per_step'), frame_segments_ms=d['frame_segments_ms']), indent=1))" > "$out/bench_segments.json"
  python tools/host_cost.py 2> /dev/null | grep "FrameRunner" > "$out/host_cost.txt"
fi
if has chain; then
  rm -f "$out/kmeans_chain_events.txt"
  for rf in "1 1 cfg2" "6 1 cfg2" "6 3 cfg2" "12 1 cfg2" "12 3 cfg2" "6 3 cfg3" "3 3 cfg4"; do
    set -- $rf
    python tools/bench_kmeans_ev.py $1 $2 5 10 $3 2> /dev/null >> "$out/kmeans_chain_events.txt"
  done
fi
if has gates; then
  python tools/bench_gates.py > "$out/gates_standalone.txt" 2> /dev/null
  python tools/bench_gates.py --config cfg4 --reps 10 > "$out/gates_standalone_cfg4.txt" 2> /dev/null
fi
if has eval; then
  python bench.py --eval-sharded --no-cpu-baseline > "$out/eval_sharded_line.json" 2> /dev/null
fi
if has ablate; then
  rm -f "$out/ablate.txt"
  for cfg in cfg2 cfg3 cfg4; do
    for a in none dense gates kmeans corr local dense,gates,kmeans; do
      AOC_ABLATE=$a python tools/ablate.py --config $cfg --python-frames --no-extras --no-cpu-baseline --exact-steps 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg without', '$a', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')" >> "$out/ablate.txt"
    done
  done
fi
cd /tmp && export TMPDIR=/tmp
if has stats; then
  for cfg in cfg2 cfg3 cfg4; do
    rm -rf /tmp/prof_stats
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --no-extras --no-cpu-baseline --exact-steps 0 > "$out/bench_${cfg}_under_rocprofv3.json" 2> /dev/null
    f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
    if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_${cfg}_bench.csv"; fi
  done
  rm -rf /tmp/prof_km
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_km -- python $GRAFT_REPO_ROOT/tools/bench_kmeans_ev.py 6 3 3 10 > /dev/null 2>&1
  f=$(find /tmp/prof_km -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_kmeans_chain_R6_F3.csv"; fi
  rm -rf /tmp/prof_eval
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_eval -- python $GRAFT_REPO_ROOT/bench.py --eval-sharded --no-cpu-baseline --eval-scale 0.015 > /dev/null 2>&1
  f=$(find /tmp/prof_eval -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_eval_sharded.csv"; fi
fi
if has pmc; then
  rm -f "$out/pmc_kmeans_R6_F3.txt"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_km
    timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_km -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_kmeans_ev.py 6 3 2 5 > /tmp/pmc_km.log 2>&1
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
fi
ls -la "$out"
