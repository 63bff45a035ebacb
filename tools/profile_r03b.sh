#!/bin/bash
# Round-3 (second half) measurement artifacts (GPU box): default bench line, rocprofv3 kernel statistics of the bench command, the
# k-means chain alone, the sequence-sharded evaluation line.  Usage: tools/profile_r03b.sh   (outputs under gpurun_out/r03b/)
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r03b
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
python bench.py > "$out/bench_line.json" 2> "$out/bench_line.err"
python bench.py --eval-sharded --no-cpu-baseline > "$out/eval_sharded_line.json" 2> /dev/null
for R in 1 6 12; do python tools/bench_kmeans.py $R 1 2>&1 | grep "k-means chain" >> "$out/kmeans_chain_standalone.txt"; done
python tools/bench_kmeans.py 6 3 2>&1 | grep "k-means chain" >> "$out/kmeans_chain_standalone.txt"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --exact-steps 0 > "$out/bench_under_rocprof.json" 2> /dev/null
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_default_bench.csv"; fi
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_eval -- python $GRAFT_REPO_ROOT/bench.py --eval-sharded --no-cpu-baseline --eval-scale 0.015 > /dev/null 2>&1
f=$(find /tmp/prof_eval -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_eval_sharded.csv"; fi
ls -la "$out"
