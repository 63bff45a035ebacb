#!/bin/bash
# Round 6: the coalesced seed kernel (eight lanes per pixel) and seeds from the newest 1 / 2 / 3 / 6 frames, in the bench.  Output: gpurun_out/r06b/bench_seed_frames.txt
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
{
python -m pytest tests/test_gpu_dense_split.py -q 2>&1 | tail -2
for rep in 1 2 3; do
for lib in libaoc_hip.so libaoc_hip_seed2.so libaoc_hip_seed3.so libaoc_hip_seed6.so; do
  for cfg in cfg2; do
  echo "== bench $cfg $lib"
  AOC_LIB_FILE=$lib python bench.py --config $cfg --no-extras --no-cpu-baseline --exact-steps 0 --details-file gpurun_out/r06b/bd_tmp.json 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], 'frames/s', d['roofline']['avg_launch_ms'], 'ms dense in-run, rescored', d['roofline'].get('rescored_pair_fraction'))"
  done
done
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --exact-steps 0 --details-file $GRAFT_REPO_ROOT/gpurun_out/r06b/bd_tmp.json > /dev/null 2>&1
f=$(find /tmp/prof_s -name "*kernel_stats.csv" | head -1); grep -i "seed\|dense_prune" "$f" | cut -d, -f1-4 | cut -c1-200
} > "$out/bench_seed_frames.txt" 2>&1
cat "$out/bench_seed_frames.txt"
