"""Developer analysis (GPU box; torch is the calculator here, nothing of the product runs), VERDICT r4 item 4b: partial-distance pruning of the dense
kernel.  For the bench's pools (frames 0, 5, 10, ... of a synthetic clip; query = last pool frame + QUERY_OFFSET) which share of the (query tile,
reference tile) pairs -- 32 query pixels x 32 reference rows of one object, the kernel's MFMA tile -- could be discarded after 2 / 3 / 4 of the 7
k-steps (16 slots each; the kernel's order: k-step 6 = channels 96..99 + norm slots first, then 0, 1, ...), i.e. EVERY one of their 1024 pairs already has an upper bound of its matching value below the query pixel's final best
value minus the kernel's rescoring margin?

value(q, r) = q.r - |r|^2 / 2 (the accumulator up to the scale 2^20; max over r <-> min distance).  After j k-steps the first 16 j channels are in:
  bound CS (Cauchy-Schwarz on the rest) : P_j + |q_rest| |r_rest| - |r|^2 / 2
  bound PD (partial distance, rest >= 0): P_j + |q_rest|^2 / 2 - |r_pre|^2 / 2         (CS <= PD always)
The "final best" is the best the kernel can ever know, so the shares are UPPER bounds of what an in-kernel test achieves.
Usage: python tools/dense_prefix_analysis.py [R=6] [cfg=cfg2]      (POOL_STRIDE=5 QUERY_OFFSET=3 by default)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aoc_amd import synthetic as syn  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
CFG = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
cfg = syn.CONFIGS[CFG]
STRIDE = int(os.environ.get("POOL_STRIDE", "5"))
QOFF = int(os.environ.get("QUERY_OFFSET", "3"))
clip = syn.make_clip(cfg, 1, frames=(R - 1) * STRIDE + QOFF + 1)
dev = "cuda" if torch.cuda.is_available() else "cpu"
emb = torch.from_numpy(clip["emb"]).to(dev)
lab = torch.from_numpy(clip["lab"]).to(dev)
hw, C, O = cfg.h * cfg.w, cfg.c, cfg.n_obj
pool = emb[0:(R - 1) * STRIDE + 1:STRIDE].reshape(-1, C)
plab = lab[0:(R - 1) * STRIDE + 1:STRIDE].reshape(-1)
q = emb[(R - 1) * STRIDE + QOFF].reshape(-1, C)
m = q.shape[0]
mq = (m + 31) // 32 * 32
qp = torch.cat([q, q[-1:].expand(mq - m, C)])                     # partial last tile: copies (a duplicate changes no statistic)
STEPS = (2, 3, 4, 5)
# the kernel's margin in value units: ~1.4e-3 in squared-distance units on these embeddings -> 0.7e-3 in value units
EPS = 0.7e-3
tot_pairs = 0
drop = {(b, j): 0 for b in ("cs", "pd", "tile") for j in STEPS}
print(f"{CFG}: R={R} pool frames (stride {STRIDE}), query = last pool frame + {QOFF}; {m} query pixels, {pool.shape[0]} pool rows, C={C}", flush=True)
for o in range(O):
    rows = torch.nonzero(plab == o).flatten().flip(0)             # the kernel lists the newest pool frame first
    n = rows.numel()
    if n == 0:
        continue
    nr = (n + 31) // 32 * 32
    r = pool[rows]
    r = torch.cat([r, r[:1].expand(nr - n, C)])                   # partial last tile: copies of the first row (as the kernel does)
    rn2 = (r * r).sum(1)
    # final best value per query pixel
    best = torch.full((mq,), -1e30, device=dev)
    QB = 2048
    for q0 in range(0, mq, QB):
        v = qp[q0:q0 + QB] @ r.T - 0.5 * rn2[None, :]
        best[q0:q0 + QB] = v.max(1).values
    thr = best - EPS
    n_rt = nr // 32
    for j in STEPS:
        # the kernel's k order: k-step 6 (channels 96..99 + the norm slots) first, then k-steps 0, 1, ... -> after j MFMA steps the channels
        # 96..99 and 0 .. 16 (j - 1) - 1 are in
        pre = torch.zeros(C, dtype=torch.bool, device=dev)
        pre[96:] = True
        pre[:16 * (j - 1)] = True
        rp, rr_ = r[:, pre], r[:, ~pre]
        r_pre2 = (rp ** 2).sum(1)
        r_rest = (rr_ ** 2).sum(1).sqrt()
        r_rest_tile = r_rest.view(n_rt, 32).max(1).values                                     # one number per reference tile
        for q0 in range(0, mq, QB):
            qq = qp[q0:q0 + QB]
            P = qq[:, pre] @ rp.T
            q_rest2 = (qq[:, ~pre] ** 2).sum(1)
            core = P - 0.5 * rn2[None, :]
            ub_cs = core + q_rest2.sqrt()[:, None] * r_rest[None, :]
            ub_pd = P + 0.5 * q_rest2[:, None] - 0.5 * r_pre2[None, :]
            thr_q = thr[q0:q0 + QB, None]
            for name, ub in (("cs", ub_cs), ("pd", ub_pd)):
                tile_max = ub.view(qq.shape[0], n_rt, 32).max(2).values                       # per query pixel, per reference tile
                dead = (tile_max < thr_q).view(-1, 32, n_rt).all(1)                           # every pixel of the query tile
                drop[(name, j)] += int(dead.sum())
            # cheap variant: max over the tile of the accumulator, plus |q_rest| x the TILE's largest |r_rest| (one FMA per lane after max16)
            tm = core.view(qq.shape[0], n_rt, 32).max(2).values + q_rest2.sqrt()[:, None] * r_rest_tile[None, :]
            drop[("tile", j)] += int((tm < thr_q).view(-1, 32, n_rt).all(1).sum())
    pairs = (mq // 32) * n_rt
    tot_pairs += pairs
    print(f"  object {o}: {n} rows = {n_rt} reference tiles x {mq // 32} query tiles", flush=True)
for name, what in (("cs", "Cauchy-Schwarz rest bound"), ("pd", "partial distance"), ("tile", "CS with the tile's max |r_rest|")):
    print(f"{what:28s}: tile pairs discardable after " + ", ".join(f"{j} k-steps {drop[(name, j)] / tot_pairs:.3f}" for j in STEPS), flush=True)
cs3 = drop[("cs", 3)] / tot_pairs
cs4 = drop[("cs", 4)] / tot_pairs
print(f"verdict bar (per-row Cauchy-Schwarz bound): >= 0.30 of the pairs after <= 3 k-steps -> {'GO' if cs3 >= 0.30 else 'NO-GO'} ({cs3:.3f}); coarse-pass MFMAs left with ONE "
      f"checkpoint (a surviving pair pays one more MFMA that takes the bound's rank-1 term out again): after 3 steps {(3 + (1 - cs3) * 5) / 7:.3f}, after 4 steps "
      f"{(4 + (1 - cs4) * 4) / 7:.3f} of today's 7 per pair", flush=True)
