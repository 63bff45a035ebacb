#!/bin/bash
# developer tool: bench.py over a list of values of one option; prints frames/s, ms per step and the in-run op averages
# usage: tools/sweep.sh --cu-reserve "32 64 96" [other bench.py options]
opt=$1; vals=$2; shift 2
for v in $vals; do
  python bench.py --no-cpu-baseline --exact-steps 0 $opt $v "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$opt', '$v', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})
"
done
