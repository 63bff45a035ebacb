#!/usr/bin/env python3
"""Stand-alone timing of every modulation gate of one frame (the 14 activations of decoding_module.py:22-84) on an idle GPU, op by op:
aoc_film_scale, aoc_plane_mean, aoc_cond_gate_pool_ex per activation shape, and the whole aoc_gates_enqueue call.

    python tools/bench_gates.py [--config cfg2] [--reps 30]

One JSON object per (op, shape): median launch time by HIP events, the bytes the op must move and the fraction of the 8 TB/s HBM peak;
last line: the sum over the frame's gates against the one-call figure."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aoc_amd  # noqa: E402
from aoc_amd import hotpath, ops  # noqa: E402
from aoc_amd import synthetic as syn  # noqa: E402
from aoc_amd.conditioning_layer import conditioning_block  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    torch.cuda._sleep(int(2e7))                   # the host enqueues the repetitions while the GPU spins: no launch gap inside an event pair
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    cfg = syn.CONFIGS[args.config]
    O, h, w = cfg.n_obj, cfg.h, cfg.w
    dev = torch.device("cuda")
    torch.manual_seed(0)
    gates = hotpath.CalibrationGates().to(dev)
    D = gates.IA1.IA.weight.shape[1]
    head = torch.randn(O, D, device=dev)
    acts = [torch.randn(O, c, hh, ww, device=dev) for _, c, hh, ww, _ in gates.plan(h, w)]

    def rotating(x):
        """Enough copies of x (and of an output) that a repetition never finds its operand in the 256 MB infinity cache."""
        n = max(2, int(640e6 // (x.numel() * 8)) + 1)
        return [x.clone() for _ in range(n)], [torch.empty_like(x) for _ in range(n)], [0]

    def cold(fn, xs, ys, it):
        def run():
            i = it[0] = (it[0] + 1) % len(xs)
            fn(xs[i], ys[i])
        return run

    tot = dict(film=0.0, mean=0.0, cond=0.0, bytes=0.0)
    with torch.no_grad():
        for (name, c, hh, ww, extra), x in zip(gates.plan(h, w), acts):
            mod = getattr(gates, name)
            n = x.numel() * 4
            xs, ys, it = rotating(x)
            if hasattr(mod, "CL_1"):
                k = int(mod.CL_1.beta_percentage * hh * ww)
                pw, pb = mod.CL_1.phi_layer.weight.reshape(-1), mod.CL_1.phi_layer.bias
                ms = timed(cold(lambda a, b: ops.cond_gate_pool(a, pw, pb, k, want_plane_mean=True), xs, ys, it), args.reps)
                tot["cond"] += ms
                tot["bytes"] += 2 * n
                print(json.dumps(dict(gate=name, op="cond_gate_pool_ex", shape=list(x.shape), ms=round(ms, 4), moved_bytes=2 * n,
                                      moved_frac=round(2 * n / (ms * 1e-3) / 8e12, 3))), flush=True)
                wt, bs = mod.mlp_layer.weight, mod.mlp_layer.bias
                hd = torch.randn(O, wt.shape[1], device=dev)
            else:
                wt, bs = mod.IA.weight, mod.IA.bias
                hd = torch.randn(O, wt.shape[1], device=dev)
                if extra:
                    ms = timed(cold(lambda a, b: ops.plane_mean(a), xs, ys, it), args.reps)
                    tot["mean"] += ms
                    tot["bytes"] += n
                    print(json.dumps(dict(gate=name, op="plane_mean", shape=list(x.shape), ms=round(ms, 4), moved_bytes=n,
                                          moved_frac=round(n / (ms * 1e-3) / 8e12, 3))), flush=True)
            ms = timed(cold(lambda a, b: ops.film_scale(a, hd, wt, bs, out=b), xs, ys, it), args.reps)
            tot["film"] += ms
            tot["bytes"] += 2 * n
            print(json.dumps(dict(gate=name, op="film_scale", shape=list(x.shape), head_dim=int(wt.shape[1]), ms=round(ms, 4), moved_bytes=2 * n,
                                  moved_frac=round(2 * n / (ms * 1e-3) / 8e12, 3))), flush=True)
            del xs, ys
        one = timed(lambda: gates.forward_batched(acts, head), args.reps)
    s = tot["film"] + tot["mean"] + tot["cond"]
    print(json.dumps(dict(frame_gates_sum_ms=round(s, 4), film_ms=round(tot["film"], 4), plane_mean_ms=round(tot["mean"], 4), cond_ms=round(tot["cond"], 4),
                          one_call_ms=round(one, 4), moved_bytes=tot["bytes"], moved_frac_one_call=round(tot["bytes"] / (one * 1e-3) / 8e12, 3))), flush=True)


if __name__ == "__main__":
    main()
