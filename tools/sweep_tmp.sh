cd /tmp && export TMPDIR=/tmp
for c in cfg3 cfg4; do
rm -rf /tmp/pc
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 20 --no-extras --no-cpu-baseline --exact-steps 0 > /tmp/b.json 2>/dev/null
f=$(find /tmp/pc -name "*kernel_stats.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/r04_kernel_stats_$c.csv
python - $f $c <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
fr=[int(r['Calls']) for r in rows if 'dense_prune' in r['Name']][0]
print(sys.argv[2],'frames',fr)
for r in rows[:22]:
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:44]
    print(f"  {n:46s} per_frame={int(r['Calls'])/fr:6.2f} avg={float(r['AverageNs'])/1e3:8.1f}us {float(r['Percentage']):5.2f}%")
PY
done
