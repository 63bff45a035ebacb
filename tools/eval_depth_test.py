#!/usr/bin/env python3
"""Developer experiment (GPU box): the closed evaluation loop with the host's run-ahead limited -- per lane (AOC_EVAL_DEPTH = frames of a lane
in flight) or over all lanes (AOC_EVAL_DEPTH_TOTAL).  python tools/eval_depth_test.py [davis17|cfg5]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aoc_amd
from aoc_amd import eval_runner
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "davis17"
specs = eval_runner.make_sequence_set(kind, scale=0.27 if kind == "davis17" else 0.12, seed=0)
np.random.seed(1234)
with torch.no_grad():
    eval_runner.eval_sharded(specs[:1], 0, 1, dev, max_frames=3)
    eval_runner.eval_sharded(specs, 0, 1, dev, lanes=4)
    for lanes, d, dt in [(4, 0, 0), (4, 1, 0), (4, 2, 0), (4, 3, 0), (4, 0, 2), (4, 0, 3), (4, 0, 4), (4, 0, 6), (6, 1, 0), (6, 0, 4), (8, 1, 0), (8, 0, 4), (8, 0, 6), (4, 0, 0), (3, 0, 0), (3, 1, 0), (2, 0, 0)]:
        os.environ["AOC_EVAL_DEPTH"], os.environ["AOC_EVAL_DEPTH_TOTAL"] = str(d), str(dt)
        tot = eval_runner.eval_sharded(specs, 0, 1, dev, lanes=lanes)
        print(f"lanes={lanes} depth per lane={d} total={dt}: {tot['frames'] / float(tot['loop_seconds_max']):.1f} frames/s  J={tot['mean_j']:.6f}", flush=True)
