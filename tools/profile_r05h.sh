#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
python tools/bench_cond_persample.py > "$out/cond_persample.txt" 2> /dev/null
python -m pytest tests/test_gpu_eval_loop.py tests/test_gpu_closed_loop.py -x -q -m gpu 2>&1 | tail -3 > "$out/eval_tests.txt"
rm -f "$out/closed_loop_ab.txt"
for b in 1 0 1 0; do
  AOC_EVAL_BLOCKING_COUNTS=$b python bench.py --eval-sharded --eval-scale 0.12 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('blocking=$b', d['value'], 'frames/s', d['eval']['frames'], 'frames', d['eval']['mean_j'], d['eval']['mean_f'])" >> "$out/closed_loop_ab.txt"
done
