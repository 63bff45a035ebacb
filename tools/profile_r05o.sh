#!/bin/bash
# Round 5: calibration of rocprofv3's FETCH_SIZE for the gathered 400 / 112 / 80-byte pieces the k-means sum kernels read (tools/probe/fetch_probe.hip).
# Output: gpurun_out/r05o/fetch_size_calibration.txt
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r05o
mkdir -p "$out"
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_probe $GRAFT_REPO_ROOT/tools/probe/fetch_probe.hip > /dev/null 2>&1 || { echo "build failed"; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_fp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_fp -o pmc --output-format csv -- /tmp/fetch_probe > /tmp/fetch_probe.log 2>&1
python3 - > "$out/fetch_size_calibration.txt" <<'PY'
import csv, glob, re
exp = [l.split() for l in open("/tmp/fetch_probe.log") if l.startswith("EXPECT")]
f = glob.glob("/tmp/pmc_fp/**/*counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == "FETCH_SIZE" and "fill_kernel" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
vals = [float(r["Counter_Value"]) / 1e3 for r in rows]            # KB -> MB
print("# FETCH_SIZE (rocprofv3 --pmc, raw, MB per dispatch, mean of 3) against the bytes each probe kernel reads: requested, distinct 64-byte sectors, distinct 128-byte lines")
print(f"# {'pattern':34s} {'FETCH_SIZE':>10s} {'requested':>10s} {'sect64':>10s} {'line128':>10s}   raw/requested  raw/sect64  raw/line128")
for i, e in enumerate(exp):
    name = " ".join(e[1:-3])
    req, s64, l128 = (float(x.split("=")[1]) for x in e[-3:])
    v = vals[3 * i:3 * i + 3]
    if len(v) < 3: break
    m = sum(v) / 3
    print(f"  {name:34s} {m:10.2f} {req:10.2f} {s64:10.2f} {l128:10.2f}   {m/req:13.3f} {m/s64:11.3f} {m/l128:12.3f}")
PY
cat "$out/fetch_size_calibration.txt"
