#!/usr/bin/env python3
"""Developer experiment (GPU box): the closed evaluation loop against the number of hardware queues the HIP runtime maps its streams onto
(GPU_MAX_HW_QUEUES, read by the runtime when the process starts; default 4).  python tools/eval_queues_test.py [davis17|cfg5] [lanes] [side stream priority]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aoc_amd
from aoc_amd import eval_runner
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "davis17"
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
if len(sys.argv) > 3:
    eval_runner.SIDE_STREAM_PRIORITY = int(sys.argv[3])
specs = eval_runner.make_sequence_set(kind, scale=0.27 if kind == "davis17" else 0.12, seed=0)
np.random.seed(1234)
with torch.no_grad():
    eval_runner.eval_sharded(specs[:1], 0, 1, dev, max_frames=3)
    for i in range(3):
        tot = eval_runner.eval_sharded(specs, 0, 1, dev, lanes=lanes)
        print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '(default)')} {kind} lanes={lanes} side priority {eval_runner.SIDE_STREAM_PRIORITY} run {i}: {tot['frames'] / float(tot['loop_seconds_max']):.1f} frames/s  J={tot['mean_j']:.6f}", flush=True)
