cd /tmp && export TMPDIR=/tmp
for rf in "1 1 cfg2" "1 3 cfg2" "1 3 cfg3"; do
  set -- $rf
  rm -rf /tmp/prof_km
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_km -- python $GRAFT_REPO_ROOT/tools/bench_kmeans_ev.py $1 $2 2 10 $3 > /dev/null 2>&1
  f=$(find /tmp/prof_km -name "*kernel_stats.csv" | head -1)
  echo "== R=$1 F=$2 $3"
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
    print(f"  {n:45s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} min_us={float(r['MinNs'])/1e3:8.1f} pct={float(r['Percentage']):6.2f}")
PY
done
