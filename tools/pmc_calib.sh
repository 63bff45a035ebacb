#!/bin/bash
# Developer tool (GPU box): FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes) of the calibration kernels of tools/bench_calib.py
# (aoc_film_scale, aoc_cond_gate_pool_ex, conditioning_block at the cfg2 activation shapes).  Per-dispatch averages in KB.
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_cal
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_cal -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_calib.py > /tmp/pmc_cal.log 2>&1
  python3 - "$ctr" <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/pmc_cal/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); print(open("/tmp/pmc_cal.log").read()[-800:]); raise SystemExit
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"]
    if r["Counter_Name"] == sys.argv[1] and ("film" in n or "cond_" in n or "plane_mean" in n):
        key = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40] + " grid=" + r.get("Grid_Size", "?")
        acc[key].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{sys.argv[1]:10s} {k:64s} n={len(v):4d} avg={sum(v)/len(v):10.1f} KB")
PY
done
