#!/bin/bash
# Round-5: what the stitch does with the speculative summaries (instrumented build), per-kernel tables of the cfg3 chain with / without them.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
bash tools/build_variant.sh kstat labels_kmeans.hip "-DAOC_KS_STATS -DAOC_DEV" > /dev/null 2>&1
rm -f "$out/spec_fold_stats.txt"
for spec in "6 3 cfg2" "12 1 cfg2" "6 3 cfg3" "3 3 cfg4"; do
  set -- $spec
  for sp in 0 1; do
    echo "== AOC_KM_SPEC=$sp $spec" >> "$out/spec_fold_stats.txt"
    AOC_LIB_FILE=libaoc_hip_kstat.so AOC_KM_SPEC=$sp python tools/bench_kmeans_ev.py $1 $2 2 5 $3 2> /dev/null | tail -2 >> "$out/spec_fold_stats.txt"
  done
done
cd /tmp && export TMPDIR=/tmp
for sp in 0 1; do
  rm -rf /tmp/prof_km
  AOC_LIB_VARIANT=dev AOC_KM_SPEC=$sp timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_km -- python $GRAFT_REPO_ROOT/tools/bench_kmeans_ev.py 6 3 2 10 cfg3 > /dev/null 2>&1
  f=$(find /tmp/prof_km -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_kmeans_chain_cfg3_R6_F3_spec$sp.csv"; fi
done
