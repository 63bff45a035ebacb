#!/usr/bin/env python3
"""Developer micro-benchmark (GPU box): aoc_proxy_corr_min alone at cfg2 sizes (the 'correlation kernel')."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import aoc_amd
from aoc_amd import ops, synthetic as syn
cfg = syn.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
hw, C, O, K = cfg.h * cfg.w, cfg.c, cfg.n_obj, cfg.k
torch.manual_seed(0)
q = (torch.relu(torch.randn(hw, C)) * 0.3).cuda()
P = O * 2 * K + O
table = (torch.relu(torch.randn(P, C)) * 0.3).cuda()
sq = (table * table).sum(1)
feat = torch.empty(O, 24, cfg.h, cfg.w, device="cuda")
sb, ss, so = [], [], []
for o in range(O):
    for f in range(2):
        sb.append((o * 2 + f) * K); ss.append(K); so.append(o * 24 * hw + (1 + f) * hw)
for o in range(O):
    sb.append(O * 2 * K + o); ss.append(1); so.append(o * 24 * hw + 3 * hw)
bias = torch.zeros(3 * O, device="cuda")
for _ in range(3):
    ops.proxy_corr_min(q, table, sq, sb, ss, so, bias, feat, 1, True)
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
n = 20
e0.record()
for _ in range(n):
    ops.proxy_corr_min(q, table, sq, sb, ss, so, bias, feat, 1, True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
byt = hw * C * 4 + P * C * 4 + hw * 3 * O * 4
print(f"{cfg.name}: proxy_corr_min {ms*1e3:.1f} us  algorithmic {byt/1e6:.2f} MB -> {byt/ms/1e6:.0f} GB/s ({100*byt/ms/1e6/8000:.1f}% of 8 TB/s);  {2.0*hw*P*C/ms/1e9:.1f} TFLOP/s fp32")
