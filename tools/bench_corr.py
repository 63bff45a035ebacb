#!/usr/bin/env python3
"""Stand-alone timing of the correlation kernel (aoc_proxy_corr_min[_batched]) on an idle GPU.

    python tools/bench_corr.py [--config cfg2|cfg3|cfg4] [--batches 1,4,16,32] [--reps 30]

Prints one JSON object per batch size: average launch time (HIP events around `reps` back-to-back launches), algorithmic bytes per
launch (m C 4 + P C 4 + 4 m n_set per frame) and the fraction of the 8 TB/s HBM roofline.  Every batch uses DISTINCT
frames (own query, proxy table and output buffers), so the launch streams batch x 11.6 MB and cannot live on cache reuse of one frame."""
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

LEVELS = {"cfg2": [16], "cfg3": [8, 16, 32], "cfg4": [64]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--batches", default="1,4,16,32")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--precision", default="split")
    ap.add_argument("--check", action="store_true", help="compare the split launch with the exact-fp32 one")
    ap.add_argument("--records", action="store_true", help="aoc_proxy_corr_min_records: the queries as tile-major split records")
    args = ap.parse_args()
    cfg = syn.CONFIGS[args.config]
    levels = LEVELS[args.config]
    mc = hotpath.MatchingConfig(CLUSTER_LEVELS=levels)
    O, C, hw = cfg.n_obj, cfg.c, cfg.h * cfg.w
    L, kmax = len(levels), max(levels)
    n_ad = L * O * 2 * kmax
    n_ch = mc.proto_channels
    ch = hotpath.channel_slices(mc)
    stride = n_ch * hw
    sb, ss, so = [], [], []
    for l, k in enumerate(levels):
        for o in range(O):
            for f in range(2):
                sb.append(((l * O + o) * 2 + f) * kmax)
                ss.append(k)
                so.append(o * stride + (ch["cluster"] + 2 * l + f) * hw)
    for o in range(O):
        if os.environ.get("AOC_BENCH_NO_SINGLES"):     # developer switch: leave the k = 1 proxies out (timing experiments)
            break
        sb.append(n_ad + o)
        ss.append(1)
        so.append(o * stride + ch["proxy"] * hw)
    n_set = len(sb)
    dev = torch.device("cuda")
    rng = np.random.RandomState(0)
    bmax = max(int(b) for b in args.batches.split(","))
    frames = []
    for i in range(bmax):
        q = torch.from_numpy(syn.fresh_embedding(rng, cfg.h, cfg.w, C)).to(dev).reshape(hw, C)
        table = torch.from_numpy((np.maximum(rng.randn(n_ad + O, C), 0) * 0.1).astype(np.float32)).to(dev)
        sqn = table.pow(2).sum(1)
        bias = torch.zeros(n_set, device=dev)
        out = torch.empty(O, n_ch, cfg.h, cfg.w, device=dev)
        frames.append((q, table, sqn, bias, out))
    splits = [ops.split_rows(f[0], tiled=True) for f in frames] if args.records else None
    algo = hw * C * 4 + (n_ad + O) * C * 4 + 4 * hw * n_set
    flops = 2.0 * hw * sum(ss) * C
    for b in [int(v) for v in args.batches.split(",")]:
        fr = frames[:b]
        if args.records:
            rfr = [(f[0], sp, f[1], f[2], f[3], f[4]) for f, sp in zip(fr, splits)]
            run = lambda: ops.proxy_corr_min_records(rfr, sb, ss, so, True)
        else:
            run = lambda: ops.proxy_corr_min_batched(fr, sb, ss, so, True, args.precision)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.reps + 1)]
        evs[0].record()
        for i in range(args.reps):
            run()
            evs[i + 1].record()
        torch.cuda.synchronize()
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.reps))
        ms = per[len(per) // 2]          # median call (the first use of fresh buffers costs tens of ms once)
        gbs = b * algo / (ms * 1e-3) / 1e9
        line = dict(config=args.config, precision="split, query as records" if args.records else args.precision, frames_per_launch=b, avg_launch_ms=round(ms, 4), us_per_frame=round(ms * 1e3 / b, 2),
                    algorithmic_bytes_per_launch=b * algo, achieved_gbs=round(gbs, 1), frac_of_8TBs=round(gbs / 8000.0, 4),
                    algorithmic_tflops=round(b * flops / (ms * 1e-3) / 1e12, 2), n_set=n_set, proxies=n_ad + O,
                    per_call_ms_min_med_max=[round(per[0], 4), round(per[len(per) // 2], 4), round(per[-1], 4)])
        line["takeover_flag"] = [int(w[:4].view(torch.int32).item()) for w in ops._corr_ws.values()]
        if args.check:
            want = [torch.empty_like(f[4]) for f in fr]
            ops.proxy_corr_min_batched([(f[0], f[1], f[2], f[3], w) for f, w in zip(fr, want)], sb, ss, so, True, "fp32")
            torch.cuda.synchronize()
            chans = sorted(set((o % stride) // hw for o in so))
            line["max_abs_diff_vs_fp32"] = max(float((f[4][:, chans] - w[:, chans]).abs().max()) for f, w in zip(fr, want))
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
