#!/bin/bash
# Round 5: kernel trace of the closed evaluation loop on the cfg2-shaped (DAVIS-17-val-like) sequences -- is the GPU busy while the host waits
# for the row counts?  Output: gpurun_out/r05n/
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r05n
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ev
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ev -- python $GRAFT_REPO_ROOT/tools/eval_hostprof.py --once --lanes ${1:-4} > "$out/eval_once.txt" 2>&1
t=$(find /tmp/prof_ev -name "*kernel_trace.csv" | head -1)
python3 - "$t" > "$out/eval_trace_busy.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]) for r in rows)
# the last run = the events after the largest gap in the second half of the trace (sequence synthesis on the host between the runs)
gaps = [(ev[i + 1][0] - max(e[1] for e in ev[max(0, i - 50):i + 1]), i) for i in range(len(ev) // 3, len(ev) - 1)]
g, i = max(gaps)
ev = ev[i + 1:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
wall = t1 - t0
pts = sorted([(s, 1) for s, e, n in ev] + [(e, -1) for s, e, n in ev])
busy = depth = 0; last = None; conc = collections.Counter()
for t, d in pts:
    if last is not None:
        conc[min(depth, 4)] += t - last
        if depth > 0: busy += t - last
    depth += d; last = t
print(f"last run: {len(ev)} kernels, wall {wall/1e6:.1f} ms, GPU busy (union) {busy/1e6:.1f} ms = {busy/wall*100:.1f} %, sum of kernel durations {sum(e-s for s,e,_ in ev)/1e6:.1f} ms")
print("time at concurrency 0/1/2/3/4+ (ms):", [round(conc[k]/1e6, 1) for k in range(5)])
per = collections.Counter(); cnt = collections.Counter()
for s, e, n in ev: per[n] += e - s; cnt[n] += 1
for n, v in per.most_common(30):
    print(f"  {n:46s} n={cnt[n]:6d} {v/1e6:8.1f} ms  {v/wall*100:5.1f} % of wall")
PY
