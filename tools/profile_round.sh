#!/bin/bash
# Developer tool (GPU box): default bench line + rocprofv3 kernel stats of the same command + PMC traffic passes.
# Usage: tools/profile_round.sh <tag>     (outputs under gpurun_out/<tag>/)
set -u
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 600 python $GRAFT_REPO_ROOT/bench.py > "$out/bench.json" 2> "$out/bench.err"
tail -c 600 "$out/bench.err"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $GRAFT_REPO_ROOT/bench.py > "$out/bench_under_rocprof.json" 2> /dev/null
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$out/kernel_stats.csv"; fi
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_$ctr -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /dev/null 2>&1
  f=$(find /tmp/prof_$ctr -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" "$ctr" > "$out/pmc_$ctr.txt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") != sys.argv[2]:
        continue
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    acc[k][0] += 1
    acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k:62s} dispatches {n:6d}  sum {v:16.1f}  per-dispatch {v / n:14.2f}")
PY
  fi
done

# ---- round 2: the correlation kernel and the calibration streams by themselves (tools/bench_corr.py, tools/bench_calib.py)
python $GRAFT_REPO_ROOT/tools/bench_corr.py --config cfg2 --batches 1,4,16,32 --check > "$out/corr_cfg2.jsonl" 2>/dev/null
python $GRAFT_REPO_ROOT/tools/bench_corr.py --config cfg4 --batches 1,4,8 --check > "$out/corr_cfg4.jsonl" 2>/dev/null
python $GRAFT_REPO_ROOT/tools/bench_corr.py --config cfg3 --batches 1,4,16 --check > "$out/corr_cfg3.jsonl" 2>/dev/null
python $GRAFT_REPO_ROOT/tools/bench_calib.py --config cfg2 > "$out/calib_cfg2.jsonl" 2>/dev/null
python $GRAFT_REPO_ROOT/tools/bench_calib.py --config cfg4 > "$out/calib_cfg4.jsonl" 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_corr -- python $GRAFT_REPO_ROOT/tools/bench_corr.py --config cfg2 --batches 16 --reps 20 > /dev/null 2>&1
f=$(find /tmp/prof_corr -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_corr_B16.csv"; fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_calib -- python $GRAFT_REPO_ROOT/tools/bench_calib.py --config cfg2 > /dev/null 2>&1
f=$(find /tmp/prof_calib -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_calib_cfg2.csv"; fi
for cfgb in "cfg2 16" "cfg4 4"; do
  set -- $cfgb
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_c_$ctr
    timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_c_$ctr -- python $GRAFT_REPO_ROOT/tools/bench_corr.py --config $1 --batches $2 --reps 5 > /dev/null 2>&1
    f=$(find /tmp/prof_c_$ctr -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then
      python - "$f" "$ctr" >> "$out/pmc_corr_$1_B$2.txt" <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if r.get("Counter_Name") == sys.argv[2] and "proxy_corr_batched" in r["Kernel_Name"]]
print(f"{sys.argv[2]} proxy_corr_batched_kernel dispatches {len(v)} per-dispatch {sum(v) / max(len(v), 1):.1f} (KB)")
PY
    fi
  done
done
$GRAFT_REPO_ROOT/tools/pmc_corr.sh SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES > "$out/pmc_sq_corr_B16.txt" 2>&1
$GRAFT_REPO_ROOT/tools/pmc_corr.sh SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS >> "$out/pmc_sq_corr_B16.txt" 2>&1

# ---- the dense kernel by itself: pools as the bench's memory policy builds them (every 5th frame), stand-alone time, share of
# rescored pairs, PMC traffic and SQ counters of dense_prune_kernel
cd $GRAFT_REPO_ROOT
for R in 1 3 6 12; do
  POOL_STRIDE=5 QUERY_OFFSET=3 timeout 200 python tools/bench_dense.py $R 2>/dev/null | grep -v "^fp32\|split_rows" >> "$out/dense_standalone.txt"
done
POOL_STRIDE=5 QUERY_OFFSET=3 tools/kstats.sh 6 python $GRAFT_REPO_ROOT/tools/bench_dense.py 6 > "$out/kernel_stats_dense_R6.txt" 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  POOL_STRIDE=5 QUERY_OFFSET=3 tools/pmc_kernel.sh dense_prune "$ctr" python $GRAFT_REPO_ROOT/tools/bench_dense.py 6 >> "$out/pmc_dense_R6.txt" 2>&1
done
POOL_STRIDE=5 QUERY_OFFSET=3 tools/pmc_kernel.sh dense_prune "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" python $GRAFT_REPO_ROOT/tools/bench_dense.py 6 >> "$out/pmc_dense_R6.txt" 2>&1
POOL_STRIDE=5 QUERY_OFFSET=3 tools/pmc_kernel.sh dense_prune "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" python $GRAFT_REPO_ROOT/tools/bench_dense.py 6 >> "$out/pmc_dense_R6.txt" 2>&1
if [ -x tools/probe/mfma_probe ]; then tools/probe/mfma_probe > "$out/mfma_probe.txt" 2>&1; fi   # built by hand (hipcc --offload-arch=gfx950 tools/probe/mfma_probe.hip); not part of the library
ls -la "$out"
