import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import aoc_amd
from aoc_amd import ops, synthetic as syn
R = int(sys.argv[1]); cfg = syn.CONFIGS["cfg2"]
clip = syn.make_clip(cfg, 0, frames=(R - 1) * 5 + 4)
emb = torch.from_numpy(clip["emb"]).cuda()
lab = torch.from_numpy(np.stack([syn.one_hot(l, cfg.n_obj) for l in clip["lab"]])).cuda()
hw, C, O = cfg.h * cfg.w, cfg.c, cfg.n_obj
pool = emb[0:(R - 1) * 5 + 1:5].reshape(-1, C).contiguous(); q = emb[(R - 1) * 5 + 3].reshape(-1, C)
labels = lab[0:(R - 1) * 5 + 1:5].reshape(-1, O).contiguous()
prep = ops.label_prep(labels); out = torch.empty(O, hw, device="cuda"); bias = torch.zeros(O, device="cuda")
ps = ops.split_rows(pool); qs = ops.split_rows(q, overflow=ps.overflow)
ops.dense_prune_stats()
ops.dense_match_min_split(q, qs, pool, ps, prep, bias, out, 1, hw, True); torch.cuda.synchronize()
st = ops.dense_prune_stats()
cnt = labels.sum(0).cpu().numpy()
tiles = np.ceil(cnt / 32)
print(f"R={R} seeds={os.environ.get('AOC_DENSE_SEED','1')}: rescored {st['rescored']} of {st['tested']} pairs = {st['rescored']/st['tested']:.3f}; by tile object 0/1/2: {st['dev_cycles']} ; reference tiles per object {tiles.astype(int).tolist()}; share of each object's pairs rescored: {[round(st['dev_cycles'][o] / (tiles[o] * 806), 3) for o in range(3)]}")
