#!/usr/bin/env python3
"""Stand-alone timing of the calibration streams (a11-a13) on an idle GPU: aoc_film_scale, aoc_cond_gate_pool_ex, conditioning_block.

    python tools/bench_calib.py [--config cfg2|cfg4]

One JSON object per op: median launch time (HIP events), algorithmic bytes (film 2 O c h w 4, gate pool O C H W 4) and the
fraction of the 8 TB/s HBM roofline; for the conditioning block also the bytes it actually moves (3 reads + 1 write of x)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aoc_amd  # noqa: E402
from aoc_amd import ops  # noqa: E402
from aoc_amd import synthetic as syn  # noqa: E402


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    args = ap.parse_args()
    cfg = syn.CONFIGS[args.config]
    O, h, w = cfg.n_obj, cfg.h, cfg.w
    dev = torch.device("cuda")
    torch.manual_seed(0)
    with torch.no_grad():
        for C in (256, 512):
            hh, ww = (h, w) if C == 256 else ((h + 1) // 2, (w + 1) // 2)
            x = torch.randn(O, C, hh, ww, device=dev)
            n = x.numel() * 4
            phi_w, phi_b = torch.randn(C, device=dev) * 0.1, torch.zeros(1, device=dev)
            k = int(0.3 * hh * ww)
            ms = timed(lambda: ops.cond_gate_pool(x, phi_w, phi_b, k, want_plane_mean=True))
            print(json.dumps(dict(op="cond_gate_pool_ex (scores + plane means + k-th largest + masked pooling)", shape=list(x.shape), ms=round(ms, 4),
                                  algorithmic_bytes=n, frac_of_8TBs=round(n / (ms * 1e-3) / 8e12, 4), moved_bytes=2 * n,
                                  moved_frac=round(2 * n / (ms * 1e-3) / 8e12, 4))), flush=True)
            head = torch.randn(O, 400, device=dev)
            wt, bs = torch.randn(C, 400, device=dev) * 0.05, torch.zeros(C, device=dev)
            ms = timed(lambda: ops.film_scale(x, head, wt, bs))
            print(json.dumps(dict(op="film_scale", shape=list(x.shape), ms=round(ms, 4), algorithmic_bytes=2 * n, frac_of_8TBs=round(2 * n / (ms * 1e-3) / 8e12, 4))),
                  flush=True)
            blk = aoc_amd.conditioning_layer.conditioning_block(C, 400, 0.3).to(dev)
            ms = timed(lambda: blk(x, head))
            print(json.dumps(dict(op="conditioning_block (a13)", shape=list(x.shape), ms=round(ms, 4), algorithmic_bytes=3 * n,
                                  note="algorithmic = a12 (one read) + a11 (read + write); moved = 3 reads + 1 write", moved_bytes=4 * n,
                                  frac_of_8TBs=round(3 * n / (ms * 1e-3) / 8e12, 4), moved_frac=round(4 * n / (ms * 1e-3) / 8e12, 4))), flush=True)


if __name__ == "__main__":
    main()
