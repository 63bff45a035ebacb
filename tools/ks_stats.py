#!/usr/bin/env python3
"""Developer tool (GPU box): builds an instrumented copy of the HIP library in /tmp (-DAOC_KS_STATS) and
prints how often the exact scan-sum k-means update takes its folded / careful / serial paths."""
import ctypes, os, subprocess, sys, glob
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "robust-video-object-segmentation_amd", "csrc")
so = "/tmp/libaoc_hip_stats.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DAOC_KS_STATS", "-shared"]
                      + sorted(glob.glob(os.path.join(csrc, "*.hip"))) + ["-o", so])
import aoc_amd
aoc_amd._lib.SO_PATH = so
L = aoc_amd._lib.lib()
from aoc_amd import synthetic as syn
cfg = syn.CONFIGS["cfg2"]
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
clip = syn.make_clip(cfg, 1, frames=5 * R + 1)
idx = [0] + [5 * i for i in range(1, R)]
emb = torch.from_numpy(clip["emb"][idx]).cuda().reshape(-1, cfg.c)
lab = torch.from_numpy(np.stack([syn.one_hot(clip["lab"][i], cfg.n_obj) for i in idx])).cuda().reshape(-1, cfg.n_obj)
np.random.seed(0)
buf = (ctypes.c_ulonglong * 8)()
L.aoc_debug_ks_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.aoc_debug_ks_stats(buf, 1)
cp = aoc_amd.matching.cluster_proxies(emb, lab)
torch.cuda.synchronize()
L.aoc_debug_ks_stats(buf, 0)
names = ["big wave: total cycles", "big wave: summary-step cycles", "big wave: chunks needing rows", "big wave: cycles waiting for rows", "big wave: exact-path cycles (incl. waiting for rows)", "big wave: chunk-exact calls", "big wave: crossings resolved", "big wave: fallback block-exact calls"]
print("R =", R, "rows", emb.shape[0], "counts", cp["counts"], "max cluster", int(cp["cluster_counts"].max()))
for n, v in zip(names, list(buf)): print(f"{n:56s} {v}")
t0 = torch.cuda.Event(True); t1 = torch.cuda.Event(True)
np.random.seed(0); t0.record(); cp = aoc_amd.matching.cluster_proxies(emb, lab); t1.record(); torch.cuda.synchronize()
print("cluster_proxies ms", t0.elapsed_time(t1))
