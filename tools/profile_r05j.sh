#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_frame.py tests/test_gpu_round4.py tests/test_gpu_eval_loop.py -x -q -m gpu 2>&1 | tail -3 > "$out/advice_tests.txt"
python -m cProfile -o /tmp/bench.prof bench.py --no-extras --no-cpu-baseline --exact-steps 0 --min-region-s 2.0 > /dev/null 2>&1
python - > "$out/host_profile_cfg2.txt" <<'PY'
import pstats
p = pstats.Stats("/tmp/bench.prof")
p.sort_stats("cumulative").print_stats(45)
p.sort_stats("tottime").print_stats(30)
PY
