#!/usr/bin/env python3
"""Developer tool (GPU box): runs aoc_kmeans_segmented a few times at cfg2 sizes for a given pool size R (for rocprofv3)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import aoc_amd
from aoc_amd import synthetic as syn, ops
cfg = syn.CONFIGS["cfg2"]
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
clip = syn.make_clip(cfg, 1, frames=5 * R + 1)
idx = [0] + [5 * i for i in range(1, R)]
emb = torch.from_numpy(clip["emb"][idx]).cuda().reshape(-1, cfg.c)
lab = torch.from_numpy(np.stack([syn.one_hot(clip["lab"][i], cfg.n_obj) for i in idx])).cuda().reshape(-1, cfg.n_obj)
prep = ops.label_prep(lab)
seg_k = ops.kmeans_plan(prep.counts, cfg.n_obj, 16)
counts = prep.counts.cpu().numpy()
rows = syn.kmeans_init_rows(3, counts[:cfg.n_obj], 16)
init = np.zeros((cfg.n_obj, 16), np.int32)
for o, r in enumerate(rows):
    init[o, :len(r)] = r
init = torch.from_numpy(init).cuda()
for _ in range(reps):
    out = ops.kmeans_segmented(emb, prep.obj_rows, prep.obj_offsets, seg_k, init, 16, 20, rows_capacity=prep.obj_rows.numel())
torch.cuda.synchronize()
print("R", R, "max cluster", int(out[2].max()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    out = ops.kmeans_segmented(emb, prep.obj_rows, prep.obj_offsets, seg_k, init, 16, 20, rows_capacity=prep.obj_rows.numel())
e1.record()
torch.cuda.synchronize()
print("R", R, "kmeans %.3f ms/call (%s)" % (e0.elapsed_time(e1) / 5, os.environ.get("AOC_KM_SUM", "ordered")))
