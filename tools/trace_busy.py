"""Developer tool: from a rocprofv3 kernel-trace CSV, print the GPU-busy union, the per-kernel share and concurrency."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, e, r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40], r.get("Queue_Id", "")))
ev.sort()
t0, t1 = ev[len(ev) // 5][0], ev[-1][1]          # skip the first fifth (warm-up)
ev = [x for x in ev if x[0] >= t0]
wall = t1 - t0
pts = []
for s, e, n, q in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = 0; depth = 0; last = None; conc = collections.Counter()
for t, d in pts:
    if last is not None and depth > 0:
        busy += t - last
    if last is not None:
        conc[min(depth, 4)] += t - last
    depth += d; last = t
print(f"wall {wall/1e6:.1f} ms, union busy {busy/1e6:.1f} ms ({busy/wall*100:.1f}%), sum of durations {sum(e-s for s,e,_,_ in ev)/1e6:.1f} ms")
print("time at concurrency 0/1/2/3/4+ (ms):", [round(conc[i]/1e6, 1) for i in range(5)])
per = collections.Counter(); perq = collections.Counter()
for s, e, n, q in ev:
    per[n] += e - s; perq[q] += e - s
for n, v in per.most_common(14):
    print(f"  {n:42s} {v/1e6:8.1f} ms  {v/wall*100:5.1f}% of wall")
print("per queue (ms):", {k: round(v/1e6, 1) for k, v in perq.items()})
