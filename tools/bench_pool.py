import sys, numpy as np, torch
sys.path.insert(0, '/root/repo' if len(sys.argv) < 2 else sys.argv[1])
import aoc_amd
from aoc_amd import ops
def timed(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0=[torch.cuda.Event(enable_timing=True) for _ in range(reps)]; e1=[torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    for i in range(reps):
        e0[i].record(); fn(); e1[i].record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a,b in zip(e0,e1)]))
hw, C, O = 121*213, 100, 4
for R in (1, 6, 12):
    emb = torch.rand(R, hw, C, device='cuda'); lab = torch.zeros(R, hw, O, device='cuda'); lab[..., 0] = 1
    ms = timed(lambda: ops.masked_mean_pool(emb, lab, 1e-5, pixel_major=True))
    pos, neg = ops.masked_mean_pool(emb, lab, 1e-5, pixel_major=True)
    print(f"masked_mean_pool R={R}: {ms*1e3:.1f} us  checksum {float(pos.double().sum()):.9f} {float(neg.double().sum()):.9f}")
