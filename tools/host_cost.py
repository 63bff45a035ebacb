#!/usr/bin/env python3
"""Developer tool (GPU box): host time of the per-frame enqueue calls -- the FrameRunner call (Python part / the C call), the gates call and a
k-means chain launch -- while the GPU drains in the background.
    python tools/host_cost.py [--pool-frames 6] [--calls 100]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aoc_amd  # noqa: E402
from aoc_amd import hotpath, ops, synthetic as syn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pool-frames", type=int, default=6)
    ap.add_argument("--calls", type=int, default=100)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = syn.CONFIGS["cfg2"]
    mc = hotpath.MatchingConfig()
    O, R = cfg.n_obj, args.pool_frames
    clip = syn.make_clip(cfg, 3, frames=R * mc.MEM_EVERY + 2)
    emb = torch.from_numpy(clip["emb"]).to(dev)
    lab = torch.from_numpy(np.stack([syn.one_hot(l, O) for l in clip["lab"]])).to(dev)
    ref_emb, ref_lab = emb[0:R * mc.MEM_EVERY:mc.MEM_EVERY].contiguous(), lab[0:R * mc.MEM_EVERY:mc.MEM_EVERY].contiguous()
    counts = [int(ref_lab[..., o].sum().item()) for o in range(O)]
    rows = np.zeros((O, 16), np.int32)
    for o, r in enumerate(syn.kmeans_init_rows(5, counts, 16)):
        rows[o, :len(r)] = r
    init = torch.from_numpy(rows).to(dev)
    runner = hotpath.FrameRunner(mc, cfg.h, cfg.w, cfg.c, O, R, dev)
    bias = torch.zeros(O, device=dev)
    tq = R * mc.MEM_EVERY + 1
    side = torch.cuda.Stream(dev, priority=-1)
    gates = hotpath.CalibrationGates().to(dev)
    acts = [torch.randn(O, c, hh, ww, device=dev) for _, c, hh, ww, _ in gates.plan(cfg.h, cfg.w)]
    L = aoc_amd._lib.lib()
    real = L.aoc_frame_enqueue
    spent = [0.0]

    class Timed:
        def __call__(self, *a):
            t0 = time.perf_counter()
            r = real(*a)
            spent[0] += time.perf_counter() - t0
            return r
    with torch.no_grad():
        ahead = hotpath.launch_cluster_proxies(mc, ref_emb, ref_lab, init, side)
        feat, head = runner(ref_emb, ref_lab, emb[tq - 1], lab[tq - 1], emb[tq], bias, ahead, pool_key=ref_emb.shape[0])
        gates.forward_batched(acts, head)
        torch.cuda.synchronize()
        L.aoc_frame_enqueue = Timed()
        t_frame = t_gates = t_chain = 0.0
        n = args.calls
        for i in range(n):
            t0 = time.perf_counter()
            feat, head = runner(ref_emb, ref_lab, emb[tq - 1], lab[tq - 1], emb[tq], bias, ahead, pool_key=ref_emb.shape[0])
            t1 = time.perf_counter()
            gates.forward_batched(acts, head)
            t2 = time.perf_counter()
            t_frame += t1 - t0
            t_gates += t2 - t1
            if i % 8 == 7:
                torch.cuda.synchronize()
        L.aoc_frame_enqueue = real
        for i in range(20):
            t0 = time.perf_counter()
            hotpath.launch_cluster_proxies(mc, ref_emb, ref_lab, init, side)
            t_chain += time.perf_counter() - t0
            torch.cuda.synchronize()
    print(f"FrameRunner call {t_frame / n * 1e3:.3f} ms (of which the C call aoc_frame_enqueue {spent[0] / n * 1e3:.3f}), gates call {t_gates / n * 1e3:.3f} ms, "
          f"one chain launch (one frame) {t_chain / 20 * 1e3:.3f} ms")


if __name__ == "__main__":
    main()
