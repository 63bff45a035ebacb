#!/bin/bash
# Round 6: carried seeds (the rows that held each (pixel, object)'s maximum in the previous frame, AOC_DENSE_CARRY) off / on: tests, then the bench with
# libaoc_hip.so (on) against libaoc_hip_nocarry.so (-DAOC_DENSE_CARRY=0), alternating runs.  Output: gpurun_out/r06b/bench_carry.txt
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
{
python -m pytest tests/test_gpu_dense_split.py tests/test_gpu_frame.py tests/test_gpu_frame_modes.py tests/test_gpu_round4.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3
for rep in 1 2 3; do
for lib in libaoc_hip_nocarry.so libaoc_hip.so; do
  for cfg in cfg2 cfg3 cfg4; do
  echo "== bench $cfg $lib"
  AOC_LIB_FILE=$lib python bench.py --config $cfg --no-extras --no-cpu-baseline --exact-steps 0 --details-file gpurun_out/r06b/bd_tmp_$cfg.json 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); dd=json.load(open('gpurun_out/r06b/bd_tmp_$cfg.json')); print(d['value'], 'frames/s', d['roofline']['avg_launch_ms'], 'ms dense in-run, rescored', dd['roofline'].get('rescored_pair_fraction'))"
  done
done
done
} > "$out/bench_carry.txt" 2>&1
cat "$out/bench_carry.txt"
