#!/bin/bash
# Developer tool (GPU box): bench throughput with parts of the step replaced by no-ops (tools/ablate.py) and with 1/3/4 sequences in flight.
for a in none dense gates local kmeans corr dense,gates dense,gates,kmeans; do
  AOC_ABLATE=$a python tools/ablate.py --python-frames --no-extras --steps 30 --no-cpu-baseline --exact-steps 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$a', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done
for s in 1 3 4; do python bench.py --steps 30 --no-cpu-baseline --exact-steps 0 --streams $s 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('streams $s', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done
