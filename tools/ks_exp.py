#!/usr/bin/env python3
"""Developer tool (GPU box): times aoc_kmeans_segmented for experimental builds of the scan-sum kernel."""
import ctypes, os, subprocess, sys, glob, importlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "robust-video-object-segmentation_amd", "csrc")
import aoc_amd
from aoc_amd import synthetic as syn, ops
cfg = syn.CONFIGS["cfg2"]
R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
clip = syn.make_clip(cfg, 1, frames=5 * R + 1)
idx = [0] + [5 * i for i in range(1, R)]
emb = torch.from_numpy(clip["emb"][idx]).cuda().reshape(-1, cfg.c)
lab = torch.from_numpy(np.stack([syn.one_hot(clip["lab"][i], cfg.n_obj) for i in idx])).cuda().reshape(-1, cfg.n_obj)
for exp in (0, 1, 2):
    so = f"/tmp/libaoc_exp{exp}.so"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-DAOC_KS_EXP={exp}", "-shared"]
                          + sorted(glob.glob(os.path.join(csrc, "*.hip"))) + ["-o", so])
    aoc_amd._lib._lib = None
    aoc_amd._lib.SO_PATH = so
    aoc_amd._lib.lib()
    prep = ops.label_prep(lab)
    seg_k = ops.kmeans_plan(prep.counts, cfg.n_obj, 16)
    counts = prep.counts.cpu().numpy()
    rows = syn.kmeans_init_rows(3, counts[:cfg.n_obj], 16)
    init = np.zeros((cfg.n_obj, 16), np.int32)
    for o, r in enumerate(rows):
        init[o, :len(r)] = r
    init = torch.from_numpy(init).cuda()
    for _ in range(2):
        out = ops.kmeans_segmented(emb, prep.obj_rows, prep.obj_offsets, seg_k, init, 16, 20, rows_capacity=prep.obj_rows.numel())
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(3):
        out = ops.kmeans_segmented(emb, prep.obj_rows, prep.obj_offsets, seg_k, init, 16, 20, rows_capacity=prep.obj_rows.numel())
    e1.record(); torch.cuda.synchronize()
    print(f"EXP={exp} R={R}: kmeans_segmented {e0.elapsed_time(e1)/3:.3f} ms  max cluster {int(out[2].max())}")
