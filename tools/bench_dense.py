"""Developer micro-benchmark: dense matching alone (fp32-exact vs split-fp16) at cfg2 size."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import aoc_amd
from aoc_amd import ops, synthetic as syn

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = syn.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "cfg2"]
STRIDE = int(os.environ.get("POOL_STRIDE", "1"))          # pool = frames 0, STRIDE, 2 STRIDE, ... (the bench's memory policy: 5)
QOFF = int(os.environ.get("QUERY_OFFSET", "1"))           # query = last pool frame + QOFF
clip = syn.make_clip(cfg, 0, frames=(R - 1) * STRIDE + QOFF + 1)
emb = torch.from_numpy(clip["emb"]).cuda()
if os.environ.get("DATA_SCALE"):                              # timing experiment: 0 = all-zero operands (no toggling in the matrix pipe), 0.001 = tiny values
    emb = emb * float(os.environ["DATA_SCALE"])
lab = torch.from_numpy(np.stack([syn.one_hot(l, cfg.n_obj) for l in clip["lab"]])).cuda()
hw, C, O = cfg.h * cfg.w, cfg.c, cfg.n_obj
pool = emb[0:(R - 1) * STRIDE + 1:STRIDE].reshape(-1, C).contiguous()
q = emb[(R - 1) * STRIDE + QOFF].reshape(-1, C)
prep = ops.label_prep(lab[0:(R - 1) * STRIDE + 1:STRIDE].reshape(-1, O).contiguous())
out = torch.empty(O, hw, device="cuda")
bias = torch.zeros(O, device="cuda")
ps = ops.split_rows(pool)
qs = ops.split_rows(q, overflow=ps.overflow)
flops = 2.0 * hw * R * hw * C
for mode in ("fp32", "split"):
    f = (lambda: ops.dense_match_min(q, pool, prep, bias, out, 1, hw, True)) if mode == "fp32" else \
        (lambda: ops.dense_match_min_split(q, qs, pool, ps, prep, bias, out, 1, hw, True))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{mode}: R={R} {ms:.3f} ms/call  {flops / ms / 1e9:.1f} fp32-equivalent TFLOP/s", flush=True)
    res = out.clone()
    if mode == "split":
        st = ops.dense_prune_stats()
        if any(st["dev_cycles"]):
            waves = 8 * ((hw + 511) // 512)
            print("  dev_cycles per wave-tile (core clocks; tiles per wave summed over the %d timed + 3 warm-up calls): %s ; tiles %d" %
                  (n, [round(c / max(st["tiles"], 1), 1) for c in st["dev_cycles"]], st["tiles"]), flush=True)
        print("  rescored %.3f of the (reference tile, query tile) pairs; %.3f of the reference tiles had a rescoring; %.3f of the pairs stopped at the checkpoint" %
              (st["rescored"] / max(st["tested"], 1), st["tiles_rescored"] / max(st["tiles"], 1), st["stopped"] / max(st["tested"], 1)), flush=True)
    if mode == "fp32":
        ref = res
print("max |split - fp32| =", float((res - ref).abs().max()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.split_rows(q, out=qs)
e1.record()
torch.cuda.synchronize()
print("split_rows(query): %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
