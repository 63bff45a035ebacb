#!/bin/bash
# Round-3 measurement artifacts (GPU box): rocprofv3 kernel statistics of the bench command, the two k-means chains alone.
# Usage: tools/profile_r03.sh   (outputs under gpurun_out/r03/)
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --exact-steps 0 > "$out/bench_under_rocprof.json" 2> /dev/null
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_default_bench.csv"; fi
for R in 1 6 12; do
  python $GRAFT_REPO_ROOT/tools/bench_kmeans.py $R 1 2>&1 | grep "k-means chain" >> "$out/kmeans_chain_standalone.txt"
  AOC_KM_CHAIN=persistent AOC_KM_PROF=101 python $GRAFT_REPO_ROOT/tools/bench_kmeans.py $R 1 2>&1 | grep -v amdgpu >> "$out/kmeans_chain_persistent.txt"
done
AOC_KM_CHAIN=persistent timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kp -- python $GRAFT_REPO_ROOT/tools/bench_kmeans.py 6 1 > /dev/null 2>&1
f=$(find /tmp/prof_kp -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_kmeans_persistent_R6.csv"; fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kl -- python $GRAFT_REPO_ROOT/tools/bench_kmeans.py 6 1 > /dev/null 2>&1
f=$(find /tmp/prof_kl -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_kmeans_launches_R6.csv"; fi
ls -la "$out"
