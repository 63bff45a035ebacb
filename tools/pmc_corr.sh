#!/bin/bash
# SQ counter pass over the batched correlation kernel (tools/bench_corr.py); prints per-dispatch averages for the kernel.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_corr
rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_corr -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_corr.py --batches 16 --reps 5 > /tmp/pmc_corr.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pmc_corr/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file", glob.glob("/tmp/pmc_corr/**/*", recursive=True)[:20]); print(open("/tmp/pmc_corr.log").read()[-2000:]); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
acc = collections.defaultdict(list)
for r in rows:
    if "proxy_corr_batched" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:32s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
PY
