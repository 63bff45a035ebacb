"""Developer tool (GPU box, development build): per-workgroup timeline of ONE dense_prune_kernel launch (AOC_DENSE_DEBUG bit 32768 records start / prologue done /
first step done / end on the 100 MHz wall clock).  Usage: AOC_LIB_VARIANT=dev AOC_DENSE_DEBUG=32768 POOL_STRIDE=5 QUERY_OFFSET=3 python tools/dense_block_timeline.py [R]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aoc_amd
from aoc_amd import ops, synthetic as syn

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = syn.CONFIGS["cfg2"]
STRIDE, QOFF = int(os.environ.get("POOL_STRIDE", "5")), int(os.environ.get("QUERY_OFFSET", "3"))
clip = syn.make_clip(cfg, 0, frames=(R - 1) * STRIDE + QOFF + 1)
emb = torch.from_numpy(clip["emb"]).cuda()
lab = torch.from_numpy(np.stack([syn.one_hot(l, cfg.n_obj) for l in clip["lab"]])).cuda()
hw, C, O = cfg.h * cfg.w, cfg.c, cfg.n_obj
pool = emb[0:(R - 1) * STRIDE + 1:STRIDE].reshape(-1, C).contiguous()
q = emb[(R - 1) * STRIDE + QOFF].reshape(-1, C)
prep = ops.label_prep(lab[0:(R - 1) * STRIDE + 1:STRIDE].reshape(-1, O).contiguous())
out = torch.empty(O, hw, device="cuda")
bias = torch.zeros(O, device="cuda")
ps = ops.split_rows(pool)
qs = ops.split_rows(q, overflow=ps.overflow)
for _ in range(3):
    ops.dense_match_min_split(q, qs, pool, ps, prep, bias, out, 1, hw, True)
torch.cuda.synchronize()
lib = aoc_amd._lib.lib()
n = 4096 * 6
host = (ctypes.c_uint64 * n)()
lib.aoc_dev_dense_block_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.aoc_dev_dense_block_times(host, n) == 0
t = np.frombuffer(host, dtype=np.uint64).reshape(4096, 6).astype(np.int64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
us = lambda x: (x - t0) / 100.0
print(f"R={R}: {len(t)} workgroups; kernel span {us(t[:, 3].max()):.1f} us")
order = np.argsort(t[:, 0])
dur, pro, first = us(t[:, 3]) - us(t[:, 0]), (t[:, 1] - t[:, 0]) / 100.0, (t[:, 2] - t[:, 1]) / 100.0
print(f"  per workgroup: duration mean {dur.mean():.1f} min {dur.min():.1f} max {dur.max():.1f} us | prologue mean {pro.mean():.1f} max {pro.max():.1f} us | first step mean {first.mean():.1f} max {first.max():.1f} us")
print(f"  tiles per wave mean {t[:, 4].mean():.1f}; rescored pairs (wave 0) mean {t[:, 5].mean():.1f} max {t[:, 5].max()}")
starts = np.sort(us(t[:, 0]))
ends = np.sort(us(t[:, 3]))
print("  start times (us): first 5", np.round(starts[:5], 1), " median", round(float(np.median(starts)), 1), " last 5", np.round(starts[-5:], 1))
print("  end times   (us): first 5", np.round(ends[:5], 1), " median", round(float(np.median(ends)), 1), " last 5", np.round(ends[-5:], 1))
late = us(t[:, 0]) > 1.0
if late.any():
    print(f"  second-round workgroups: {late.sum()}, duration mean {dur[late].mean():.1f} us; first-round: {(~late).sum()}, duration mean {dur[~late].mean():.1f} us")
per_tile = dur / np.maximum(t[:, 4], 1)
print(f"  us per tile: mean {per_tile.mean():.3f}")
