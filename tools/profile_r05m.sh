#!/bin/bash
# Round-5: schedule knobs re-swept on the final code (the chains got shorter): chain plan, reserved CUs, literal head chunks, chains per batch.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
rm -f "$out/sweep_final.txt"
line() { python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')"; }
for plan in "2,3" "5" "1,4" "3,2" "1,2,2" "2,3"; do
  python bench.py --chain-plan $plan --no-extras --no-cpu-baseline --exact-steps 0 2> /dev/null | line "cfg2 chain-plan $plan" >> "$out/sweep_final.txt"
done
for cu in 16 24 40 48; do
  python bench.py --cu-reserve $cu --no-extras --no-cpu-baseline --exact-steps 0 2> /dev/null | line "cfg2 cu-reserve $cu" >> "$out/sweep_final.txt"
done
for hc in 16 24 28; do
  AOC_LIB_VARIANT=dev AOC_KM_HEAD_CHUNKS=$hc python bench.py --no-extras --no-cpu-baseline --exact-steps 0 2> /dev/null | line "cfg2 head chunks $hc" >> "$out/sweep_final.txt"
  AOC_LIB_VARIANT=dev AOC_KM_HEAD_CHUNKS=$hc python bench.py --config cfg3 --no-extras --no-cpu-baseline --exact-steps 0 2> /dev/null | line "cfg3 head chunks $hc" >> "$out/sweep_final.txt"
done
for cfg in cfg3 cfg4; do
  for ch in 2 3 4 5; do
    python bench.py --config $cfg --chains $ch --no-extras --no-cpu-baseline --exact-steps 0 2> /dev/null | line "$cfg chains $ch" >> "$out/sweep_final.txt"
  done
  for plan in "2,3" "1,4"; do
    python bench.py --config $cfg --chain-plan $plan --no-extras --no-cpu-baseline --exact-steps 0 2> /dev/null | line "$cfg chain-plan $plan" >> "$out/sweep_final.txt"
  done
  for cu in 16 48; do
    python bench.py --config $cfg --cu-reserve $cu --no-extras --no-cpu-baseline --exact-steps 0 2> /dev/null | line "$cfg cu-reserve $cu" >> "$out/sweep_final.txt"
  done
done
