#!/bin/bash
# Developer tool (GPU box): one rocprofv3 --pmc pass (counters only, with the kernel trace) over a command; prints the per-dispatch
# averages of every collected counter for the kernels whose name contains <pattern>.
# Usage: tools/pmc_kernel.sh <pattern> "<counters>" <command...>
pat=$1; ctrs=$2; shift 2
export PYTHONPATH=${GRAFT_REPO_ROOT:-$PWD}:${PYTHONPATH:-}
args=()
for a in "$@"; do if [ -f "$a" ]; then a=$(realpath "$a"); fi; args+=("$a"); done      # relative script paths survive the cd below
set -- "${args[@]}"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_k
rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_k -o pmc --output-format csv -- "$@" > /tmp/pmc_k.log 2>&1
python3 - "$pat" <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/pmc_k/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); print(open("/tmp/pmc_k.log").read()[-1500:]); raise SystemExit
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if sys.argv[1] in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:32s} n={len(v):3d} avg={sum(v)/len(v):.5g}")
PY
