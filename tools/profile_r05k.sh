#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_corr_batched.py tests/test_gpu_frame.py tests/test_gpu_frame_modes.py tests/test_gpu_round2.py tests/test_gpu_round4.py -x -q -m gpu 2>&1 | tail -5 > "$out/corr_multi_tests.txt"
rm -f "$out/corr_multi_bench.txt"
line() { python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d.get('roofline_correlation_kernel') or {}; print('$1', d['value'], 'frames/s', d['ms_per_step'], 'ms/step host', d.get('host_enqueue_ms_per_step'), 'corr in-run ms', c.get('avg_launch_ms'))"; }
for cfg in cfg3 cfg4 cfg2; do
  for r in 1 2; do
    python bench.py --config $cfg --no-extras --no-cpu-baseline --exact-steps 0 2> /dev/null | line "$cfg frame call (one correlation launch)" >> "$out/corr_multi_bench.txt"
    python bench.py --config $cfg --python-frames --no-extras --no-cpu-baseline --exact-steps 0 2> /dev/null | line "$cfg python frames (pass by pass)" >> "$out/corr_multi_bench.txt"
  done
done
