#!/bin/bash
# Round-5: per-kernel statistics of the cfg3 / cfg4 bench (rocprofv3 --kernel-trace --stats), the cfg4 / cfg3 step ablation, the new host metric.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for cfg in cfg3 cfg4; do
  rm -rf /tmp/prof_stats
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --no-extras --no-cpu-baseline --exact-steps 0 > "$out/bench_${cfg}_under_rocprof.json" 2> /dev/null
  f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" "$out/kernel_stats_$cfg.csv"; fi
done
cd $GRAFT_REPO_ROOT
rm -f "$out/ablate_cfg34.txt"
for cfg in cfg4 cfg3; do
  for a in none dense gates kmeans corr local dense,gates,kmeans; do
    AOC_ABLATE=$a python tools/ablate.py --config $cfg --python-frames --no-extras --no-cpu-baseline --exact-steps 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg without', '$a', d['value'], 'frames/s', d['ms_per_step'], 'ms/step host', d['host_enqueue_ms_per_step'], d.get('host_enqueue_wall_ms_per_step'))" >> "$out/ablate_cfg34.txt"
  done
done
python bench.py --no-extras --no-cpu-baseline --exact-steps 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('cfg2 frame call', d['value'], 'frames/s', d['ms_per_step'], 'ms/step host', d['host_enqueue_ms_per_step'], d.get('host_enqueue_wall_ms_per_step'))" >> "$out/ablate_cfg34.txt"
