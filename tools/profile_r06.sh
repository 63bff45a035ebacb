#!/bin/bash
# Round-6 measurement artifacts (GPU box), final code: driver-style bench line (+ --steps 20, + segments), rocprofv3 kernel statistics of the cfg2 /
# cfg3 / cfg4 bench and of the k-means chain alone, FETCH / WRITE counter passes of the chain (separate --pmc runs), hipEvent-timed chains, gates
# alone, the sequence-sharded evaluation line, the step ablations.
# Usage: tools/profile_r06.sh [part ...]   (parts: bench stats pmc pmcdense chain gates eval cfg5 ablate; default all; outputs under gpurun_out/r06f/)
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06f
mkdir -p "$out"
parts="${*:-bench stats pmc pmcdense chain gates eval timeline seedab cfg5 ablate}"
has() { case " $parts " in *" $1 "*) return 0;; *) return 1;; esac; }
cd $GRAFT_REPO_ROOT
if has bench; then
  # the driver's command, then the same with every optional leg (closed loop, backbone, correlation sweep; no wall-time budget)
  ( time python bench.py --gpus 1 --steps 20 --warmup 5 --details-file gpurun_out/r06f/bench_details_driver_style.json ) > "$out/bench_line_driver_style.json" 2> "$out/bench_line_driver_style.err"
  python bench.py --details-file gpurun_out/r06f/bench_details.json > "$out/bench_line.json" 2> "$out/bench_line.err"
  python bench.py --extras all --budget-s 0 --details-file gpurun_out/r06f/bench_details_all_legs.json > "$out/bench_line_all_legs.json" 2> /dev/null
  python bench.py --segments --no-extras --no-cpu-baseline --exact-steps 0 --details-file gpurun_out/r06f/bench_details_segments.json > /dev/null 2>&1
  python tools/host_cost.py 2> /dev/null | grep "FrameRunner" > "$out/host_cost.txt"
fi
if has pmcdense; then
  # the dense kernel alone at R = 6, bench pools: timing, then one rocprofv3 --pmc pass per counter group (separate runs), in the format bench.py parses
  export POOL_STRIDE=5 QUERY_OFFSET=3
  {
    echo "# Dense kernel alone at R = 6 (cfg2, pool = every 5th frame as in the bench): tools/bench_dense.py 6 with POOL_STRIDE=5 QUERY_OFFSET=3, then one"
    echo "# rocprofv3 --kernel-trace --pmc <counter(s)> pass per line group (tools/pmc_kernel.sh dense_prune ...; separate runs).  Per-dispatch averages;"
    echo "# FETCH_SIZE / WRITE_SIZE in KB, on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes (x2 for bytes).  Round 6, final code."
    python tools/bench_dense.py 6 2>/dev/null
    for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT"; do
      echo "--pmc $set"
      tools/pmc_kernel.sh dense_prune "$set" python tools/bench_dense.py 6
    done
  } > "$out/pmc_dense_R6.txt" 2>&1
  unset POOL_STRIDE QUERY_OFFSET
fi
if has timeline; then
  # per-workgroup wall-clock stamps of one dense launch (development build)
  ( export POOL_STRIDE=5 QUERY_OFFSET=3 AOC_LIB_VARIANT=dev AOC_DENSE_DEBUG=32768; for R in 1 6; do python tools/dense_block_timeline.py $R; done; AOC_DENSE_ROUNDS=1 python tools/dense_block_timeline.py 1 ) 2>/dev/null > "$out/dense_block_timeline.txt"
fi
if has seedab; then
  # bound seeds off / on in the bench (development build's switch), alternating runs
  rm -f "$out/bench_seed_ab.txt"
  for rep in 1 2 3; do for seed in 0 1; do for cfg in cfg2 cfg3 cfg4; do
    AOC_LIB_VARIANT=dev AOC_DENSE_SEED=$seed python bench.py --config $cfg --no-extras --no-cpu-baseline --exact-steps 0 --details-file gpurun_out/r06f/bd_tmp.json 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg development build AOC_DENSE_SEED=$seed', d['value'], 'frames/s', d['roofline']['avg_launch_ms'], 'ms dense kernel in-run')" >> "$out/bench_seed_ab.txt"
  done; done; done
fi
if has cfg5; then
  # BASELINE.json configs[4] on its own terms as far as one GPU goes: a quarter of the 30 + 507 sequence set, closed loop, one rank (RCCL forced on)
  AOC_DIST_FORCE=1 python bench.py --eval-sharded --eval-scale 0.25 --no-cpu-baseline > "$out/eval_sharded_quarter_scale_rccl1.json" 2> "$out/eval_sharded_quarter_scale.err"
fi
if has chain; then
  rm -f "$out/kmeans_chain_events.txt"
  for rf in "1 1 cfg2" "6 1 cfg2" "6 3 cfg2" "12 1 cfg2" "12 3 cfg2" "6 3 cfg3" "3 3 cfg4"; do
    set -- $rf
    python tools/bench_kmeans_ev.py $1 $2 5 10 $3 2> /dev/null >> "$out/kmeans_chain_events.txt"
  done
fi
if has gates; then
  python tools/bench_gates.py > "$out/gates_standalone.txt" 2> /dev/null
  python tools/bench_gates.py --config cfg4 --reps 10 > "$out/gates_standalone_cfg4.txt" 2> /dev/null
fi
if has eval; then
  python bench.py --eval-sharded --no-cpu-baseline > "$out/eval_sharded_line.json" 2> /dev/null
fi
if has ablate; then
  rm -f "$out/ablate.txt"
  for cfg in cfg2 cfg3 cfg4; do
    for a in none dense gates kmeans corr local dense,gates,kmeans; do
      AOC_ABLATE=$a python tools/ablate.py --config $cfg --python-frames --no-extras --no-cpu-baseline --exact-steps 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg without', '$a', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')" >> "$out/ablate.txt"
    done
  done
fi
cd /tmp && export TMPDIR=/tmp
if has stats; then
  for cfg in cfg2 cfg3 cfg4; do
    rm -rf /tmp/prof_stats
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --no-extras --no-cpu-baseline --exact-steps 0 > "$out/bench_${cfg}_under_rocprofv3.json" 2> /dev/null
    f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
    if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_${cfg}_bench.csv"; fi
  done
  rm -rf /tmp/prof_km
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_km -- python $GRAFT_REPO_ROOT/tools/bench_kmeans_ev.py 6 3 3 10 > /dev/null 2>&1
  f=$(find /tmp/prof_km -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_kmeans_chain_R6_F3.csv"; fi
  rm -rf /tmp/prof_eval
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_eval -- python $GRAFT_REPO_ROOT/bench.py --eval-sharded --no-cpu-baseline --eval-scale 0.015 > /dev/null 2>&1
  f=$(find /tmp/prof_eval -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_eval_sharded.csv"; fi
fi
if has pmc; then
  rm -f "$out/pmc_kmeans_R6_F3.txt"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_km
    timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_km -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_kmeans_ev.py 6 3 2 5 > /tmp/pmc_km.log 2>&1
    python3 - "$ctr" >> "$out/pmc_kmeans_R6_F3.txt" <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/pmc_km/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); raise SystemExit
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"]
    if "km_" in n and r["Counter_Name"] == sys.argv[1]:
        acc[n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{sys.argv[1]:10s} {k:46s} n={len(v):4d} avg={sum(v)/len(v):10.1f} KB")
PY
  done
fi
ls -la "$out"
