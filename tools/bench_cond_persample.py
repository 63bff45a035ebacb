#!/usr/bin/env python3
"""Developer experiment (GPU box), VERDICT r4 item 6: is the conditioning gate's SECOND read of z (masked pooling, after the exact k-th-largest
selection) served on-die?  aoc_cond_gate_pool_ex for all N samples in one call (scores of every sample, selection, masked pooling of every sample:
the second read of sample o comes N samples after its first) against N calls of ONE sample each (scores -> selection -> pooling of sample o back to
back: 26 MB per sample at cfg2, 59 MB at cfg4, against 256 MB of Infinity Cache).  The operand rotates through > 640 MB of copies, so nothing is
left over from the repetition before.  TIME decides (Infinity-Cache hits are counted by FETCH_SIZE).

    python tools/bench_cond_persample.py [--reps 20]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aoc_amd  # noqa: E402
from aoc_amd import ops  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    torch.cuda._sleep(int(2e7))
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    for name, shape in (("cfg2 CLB2/3", (4, 256, 121, 213)), ("cfg2 CLB4/5", (4, 512, 61, 107)), ("cfg3 CLB2/3", (6, 256, 145, 261)),
                        ("cfg4 CLB2/3", (9, 256, 181, 321)), ("cfg4 CLB4/5", (9, 512, 91, 161))):
        N, C, H, W = shape
        nbytes = N * C * H * W * 4
        n_copies = max(2, int(640e6 // nbytes) + 1)
        zs = [torch.randn(*shape, device=dev) for _ in range(n_copies)]
        phi_w, phi_b = torch.randn(C, device=dev) * 0.05, torch.zeros(1, device=dev)
        k = int(0.3 * H * W)
        it = [0]

        def whole():
            it[0] = (it[0] + 1) % n_copies
            return ops.cond_gate_pool(zs[it[0]], phi_w, phi_b, k, want_plane_mean=True)

        def per_sample():
            it[0] = (it[0] + 1) % n_copies
            z = zs[it[0]]
            return [ops.cond_gate_pool(z[o:o + 1], phi_w, phi_b, k, want_plane_mean=True) for o in range(N)]

        a = whole()
        b = per_sample()
        same = all(torch.equal(a[0][o], b[o][0][0]) and torch.equal(a[1][o], b[o][1][0]) for o in range(N))
        t_whole = timed(whole, args.reps)
        t_ps = timed(per_sample, args.reps)
        print(json.dumps(dict(shape=name, dims=list(shape), z_mb=round(nbytes / 1e6, 1), one_call_ms=round(t_whole, 4), per_sample_calls_ms=round(t_ps, 4),
                              one_call_frac_of_hbm_one_read=round(nbytes / (t_whole * 1e-3) / 8e12, 3),
                              per_sample_frac_of_hbm_one_read=round(nbytes / (t_ps * 1e-3) / 8e12, 3), identical_results=bool(same))), flush=True)
        del zs


if __name__ == "__main__":
    main()
