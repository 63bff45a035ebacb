#!/usr/bin/env python3
"""Developer micro-benchmark (GPU box): aoc_dense_match_min alone at cfg2 sizes for several pool sizes."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import aoc_amd
from aoc_amd import ops, synthetic as syn
cfg = syn.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
hw, C, O = cfg.h * cfg.w, cfg.c, cfg.n_obj
torch.manual_seed(0)
q = (torch.relu(torch.randn(hw, C)) * 0.3).cuda()
for R in (1, 3, 6, 12):
    pool = (torch.relu(torch.randn(R * hw, C)) * 0.3).cuda()
    lab = torch.zeros(R * hw, O)
    lab[torch.arange(R * hw), torch.randint(0, O, (R * hw,))] = 1
    prep = ops.label_prep(lab.cuda())
    out = torch.empty(O, hw, device="cuda")
    bias = torch.zeros(O, device="cuda")
    for _ in range(2):
        ops.dense_match_min(q, pool, prep, bias, out, 1, hw, True)
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    n = 5
    e0.record()
    for _ in range(n):
        ops.dense_match_min(q, pool, prep, bias, out, 1, hw, True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"R={R:2d} n={R*hw:7d}  {ms:8.3f} ms  {2.0*hw*R*hw*C/ms/1e9:7.1f} TFLOP/s  ({100*2.0*hw*R*hw*C/ms/1e9/157.3:.1f}% of fp32 MFMA peak)")
