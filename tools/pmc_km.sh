#!/bin/bash
# Developer tool (GPU box): FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes, counters only + kernel trace) of the k-means kernels of
# tools/bench_kmeans.py <R> <frames per chain>, with and without the XCD-aware workgroup ids (AOC_KM_XCD=0).  Per-dispatch averages in KB.
# Usage: tools/pmc_km.sh [R] [frames]
R=${1:-6}; F=${2:-3}
cd /tmp && export TMPDIR=/tmp
for xcd in 1 0; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_km
    AOC_LIB_VARIANT=dev AOC_KM_XCD=$xcd rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_km -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_kmeans.py $R $F > /tmp/pmc_km.log 2>&1
    python3 - "$xcd" "$ctr" <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/pmc_km/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); print(open("/tmp/pmc_km.log").read()[-800:]); raise SystemExit
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"]
    if "km_" in n and r["Counter_Name"] == sys.argv[2]:
        acc[n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"AOC_KM_XCD={sys.argv[1]} {sys.argv[2]:10s} {k:46s} n={len(v):4d} avg={sum(v)/len(v):10.1f} KB")
PY
  done
done
