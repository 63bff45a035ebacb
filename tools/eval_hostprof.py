#!/usr/bin/env python3
"""Developer tool (GPU box): where the host thread of the closed evaluation loop spends its time -- cProfile over eval_runner.eval_sharded on the
cfg2-shaped (DAVIS-17-val-like) part of the set, next to the loop's wall time with and without the profiler.
    python tools/eval_hostprof.py [--lanes 4] [--scale 0.27] [--out gpurun_out/eval_hostprof.txt]"""
import argparse
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aoc_amd  # noqa: E402
from aoc_amd import eval_runner  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=4)
    ap.add_argument("--scale", type=float, default=0.27)
    ap.add_argument("--set", default="davis17")
    ap.add_argument("--out", default="")
    ap.add_argument("--once", action="store_true", help="one warm run and one run at --lanes only (for a rocprofv3 kernel trace)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    specs = eval_runner.make_sequence_set(args.set, scale=args.scale, seed=0)
    np.random.seed(1234)
    out = open(args.out, "w") if args.out else sys.stdout
    with torch.no_grad():
        eval_runner.eval_sharded(specs[:1], 0, 1, dev, max_frames=3)
        if args.once:
            for _ in range(2):
                tot = eval_runner.eval_sharded(specs, 0, 1, dev, lanes=args.lanes)
                print(f"lanes={args.lanes}: {tot['frames'] / float(tot['loop_seconds_max']):.1f} frames/s", file=out, flush=True)
            return
        for lanes, stagger in [(1, 0), (2, 0), (2, 1), (2, 2), (4, 0), (4, 1), (4, 2), (4, 0), (4, 1), (6, 1), (8, 0), (8, 1), (8, 2)]:
            tot = eval_runner.eval_sharded(specs, 0, 1, dev, lanes=lanes, stagger=stagger)
            print(f"lanes={lanes} stagger={stagger}: {tot['frames'] / float(tot['loop_seconds_max']):.1f} frames/s ({int(tot['frames'])} frames, {float(tot['loop_seconds_max']):.3f} s)", file=out, flush=True)
        prof = cProfile.Profile()
        prof.enable()
        tot = eval_runner.eval_sharded(specs, 0, 1, dev, lanes=args.lanes)
        prof.disable()
        print(f"under cProfile, lanes={args.lanes}: {tot['frames'] / float(tot['loop_seconds_max']):.1f} frames/s ({float(tot['loop_seconds_max']):.3f} s for {int(tot['frames'])} frames)", file=out)
        pstats.Stats(prof, stream=out).sort_stats("cumulative").print_stats(60)
        pstats.Stats(prof, stream=out).sort_stats("tottime").print_stats(40)


if __name__ == "__main__":
    main()
