#!/bin/bash
# Round-5, first measurement pass (GPU box): (1) is the k-means chain bimodal at R = 9 / 12?  hipEvent-timed chains (tools/bench_kmeans_ev.py,
# 5 repeats x 10 chains), the old wall-clock tool three times, and rocprofv3 per-kernel tables of the chain alone; (2) probes: hipGraph host /
# GPU cost for a 105-kernel chain, legacy K = 8 f16 MFMA pass count.
# Usage: tools/profile_r05a.sh   (outputs under gpurun_out/r05/)
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -o /tmp/graph_probe tools/probe/graph_probe.hip > /dev/null 2>&1 && /tmp/graph_probe 105 2000 > "$out/graph_probe.txt" 2>&1
/tmp/graph_probe 105 20000 >> "$out/graph_probe.txt" 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_k8_probe tools/probe/mfma_k8_probe.hip > /dev/null 2>&1 && /tmp/mfma_k8_probe > "$out/mfma_k8_probe.txt" 2>&1
rm -f "$out/kmeans_chain_events.txt" "$out/kmeans_chain_wallclock.txt"
for rf in "1 1" "6 1" "6 3" "9 1" "9 3" "12 1" "12 3"; do
  python tools/bench_kmeans_ev.py $rf 5 10 2> /dev/null >> "$out/kmeans_chain_events.txt"
done
for i in 1 2 3; do python tools/bench_kmeans.py 12 1 2>&1 | grep "k-means chain" >> "$out/kmeans_chain_wallclock.txt"; done
python tools/bench_kmeans.py 6 3 2>&1 | grep "k-means chain" >> "$out/kmeans_chain_wallclock.txt"
cd /tmp && export TMPDIR=/tmp
for rf in "12 1" "9 3" "6 3"; do
  tag=$(echo $rf | tr ' ' '_')
  rm -rf /tmp/prof_km
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_km -- python $GRAFT_REPO_ROOT/tools/bench_kmeans_ev.py $rf 3 10 > "$out/kmeans_chain_events_under_rocprof_$tag.txt" 2>&1
  f=$(find /tmp/prof_km -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_kmeans_chain_R${tag}.csv"; fi
  t=$(find /tmp/prof_km -name "*kernel_trace.csv" | head -1)
  if [ -n "$t" ] && [ "$tag" = "12_1" ]; then
    python3 - "$t" > "$out/kmeans_chain_R12_F1_per_call_spread.txt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    acc[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# per-kernel duration spread over all dispatches (us): n, min, p10, median, p90, max")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    q = lambda p: v[min(len(v) - 1, int(p * len(v)))]
    print(f"{k:46s} n={len(v):5d} min={v[0]:8.1f} p10={q(0.1):8.1f} med={q(0.5):8.1f} p90={q(0.9):8.1f} max={v[-1]:8.1f} total_ms={sum(v) / 1e3:8.2f}")
PY
  fi
done
