#!/usr/bin/env python3
"""Stand-alone timing of local (windowed) matching on an idle GPU at the half-resolution map of a config (AEM:938-941):

    python tools/bench_local.py [--config cfg2] [--kernels reg,row,block]

One JSON object per kernel variant (each in a child process: the variant is a library-level developer switch, AOC_LOCAL_KERNEL, read
once): median launch time (HIP events), flops 2 m (2R+1)^2 C and the max |difference| of the outputs against the first variant."""
import argparse
import json
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))


def child(config, dump):
    import aoc_amd  # noqa: F401
    from aoc_amd import ops
    from aoc_amd import synthetic as syn
    cfg = syn.CONFIGS[config]
    clip = syn.make_clip(cfg, seed=3, frames=2)
    dev = torch.device("cuda")
    O = cfg.n_obj
    prev = torch.from_numpy(clip["emb"][0]).to(dev)
    cur = torch.from_numpy(clip["emb"][1]).to(dev)
    lab = torch.from_numpy(syn.one_hot(clip["lab"][0], O)).to(dev)
    H2, W2 = int(cfg.h / 2) + 1, int(cfg.w / 2) + 1
    q2 = ops.resize_bilinear_hwc(cur, H2, W2)
    p2 = ops.resize_bilinear_hwc(prev, H2, W2)
    bits, _ = ops.label_bits(lab.reshape(-1, O), want_wrong=False)
    bits2 = ops.resize_nearest_bits(bits, cfg.h, cfg.w, H2, W2)
    radii = [2, 4, 6, 8, 10, 12]
    bias = torch.zeros(O, device=dev)
    out = ops.local_window_match(q2, p2, bits2, radii, bias, O, True)
    ms = timed(lambda: ops.local_window_match(q2, p2, bits2, radii, bias, O, True))
    np.save(dump, out.cpu().numpy())
    flops = 2.0 * H2 * W2 * 25 * 25 * cfg.c
    print(json.dumps(dict(kernel=os.environ.get("AOC_LOCAL_KERNEL", "reg"), map=[H2, W2], ms=round(ms, 4), tflops=round(flops / ms * 1e-9, 2))), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--kernels", default="reg,row,block")
    ap.add_argument("--child", default="")
    args = ap.parse_args()
    if args.child:
        child(args.config, args.child)
        return
    ref = None
    for k in args.kernels.split(","):
        env = dict(os.environ)
        env["AOC_LOCAL_KERNEL"] = k
        env["AOC_LIB_VARIANT"] = "dev"            # library-level switches only exist in the development build
        dump = "/tmp/bench_local_%s.npy" % k
        subprocess.run([sys.executable, os.path.abspath(__file__), "--config", args.config, "--child", dump], env=env, check=True)
        o = np.load(dump)
        if ref is None:
            ref = o
        else:
            print(json.dumps(dict(kernel=k, max_abs_diff_vs_first=float(np.abs(o - ref).max()))), flush=True)


if __name__ == "__main__":
    main()
