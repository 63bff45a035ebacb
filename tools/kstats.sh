#!/bin/bash
# Developer tool (GPU box): rocprofv3 kernel stats of a command, top kernels as "name calls avg_us total_ms".
# Usage: tools/kstats.sh <n_rows> <command...>
n=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kstats_out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats_out -- "$@" > /dev/null 2>&1
f=$(find /tmp/kstats_out -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$n" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2])]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    print(f"{name:50s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e3:10.1f} us  total {float(r['TotalDurationNs']) / 1e6:9.2f} ms  {float(r['Percentage']):5.1f}%")
PY
