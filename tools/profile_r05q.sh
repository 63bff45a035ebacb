#!/bin/bash
# Round 5: schedule sweeps repeated at 16 hardware queues (the package's default since the closed-loop finding; rounds 3-5 swept at the runtime's 4):
# orchestrated bench with more sequences / chain streams in flight, closed loop lanes x stagger.  Output: gpurun_out/r05q/
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r05q
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
rm -f "$out/bench_sweep.txt"
run() { python bench.py --no-extras --no-cpu-baseline --exact-steps 0 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$*', d['value'], 'frames/s', d['ms_per_step'], 'ms/step, host', d['host_enqueue_ms_per_step'])" >> "$out/bench_sweep.txt"; }
run --config cfg2
run --config cfg2 --streams 3
run --config cfg2 --streams 4
run --config cfg2 --chain-streams 2
run --config cfg2 --streams 3 --chain-streams 2
run --config cfg3
run --config cfg3 --streams 3
run --config cfg3 --chain-streams 2
run --config cfg4
run --config cfg4 --streams 3
run --config cfg4 --chain-streams 2
GPU_MAX_HW_QUEUES=4 run --config cfg2 --streams 3
python tools/eval_hostprof.py --out "$out/eval_lanes_stagger.txt" > /dev/null 2>&1
head -14 "$out/eval_lanes_stagger.txt" > "$out/eval_lanes_stagger_head.txt"
for plan in 1 1,2,2 2,3 5 1,4; do
  AOC_EVAL_CHAIN_PLAN=$plan python tools/eval_queues_test.py cfg5 4 2>&1 | grep GPU_MAX | sed "s/^/chain plan $plan: /" >> "$out/eval_chain_plan.txt"
done
