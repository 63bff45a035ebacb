#!/usr/bin/env python3
"""Developer experiment (GPU box), VERDICT r4 item 6: does a re-read of an activation that was streamed a few tens of microseconds earlier come from
the Infinity Cache?  The operand rotates through > 640 MB of copies; every copy goes through aoc_cond_gate_pool_ex TWICE in a row.  Under
rocprofv3 --kernel-trace the durations of the odd (cold: the first read of z comes from HBM) and even (warm: z was read twice just before) calls of
cond_scores_part_kernel (first read of z) and cond_masked_gap_fused_kernel (second read of z, ~3 launches after the first) tell:
  warm scores << cold scores   -> a working set of that size IS retained on-die, so the product's second read (gap pass) already is on-die too;
  warm == cold                 -> it is not, the second read costs HBM time.
Usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/bench_cond_warm.py N C H W ; then python tools/bench_cond_warm.py --report DIR"""
import csv
import glob
import os
import sys

if sys.argv[1] == "--report":
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for kern in ("cond_scores_part_kernel", "cond_masked_gap_fused_kernel"):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if kern in r["Kernel_Name"]]
        d = d[12:]                                   # warm-up calls
        cold, warm = sorted(d[0::2]), sorted(d[1::2])
        med = lambda v: v[len(v) // 2] if v else float("nan")
        print(f"  {kern:32s} cold (first call on a copy) median {med(cold):8.1f} us   warm (second call on the same copy) median {med(warm):8.1f} us   n = {len(cold)} + {len(warm)}")
    sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aoc_amd  # noqa: E402
from aoc_amd import ops  # noqa: E402

N, C, H, W = (int(x) for x in sys.argv[1:5])
dev = torch.device("cuda")
torch.manual_seed(0)
nbytes = N * C * H * W * 4
zs = [torch.randn(N, C, H, W, device=dev) for _ in range(max(2, int(640e6 // nbytes) + 1))]
phi_w, phi_b = torch.randn(C, device=dev) * 0.05, torch.zeros(1, device=dev)
k = int(0.3 * H * W)
for it in range(6 + 40):
    z = zs[it % len(zs)]
    ops.cond_gate_pool(z, phi_w, phi_b, k, want_plane_mean=True)
    ops.cond_gate_pool(z, phi_w, phi_b, k, want_plane_mean=True)
torch.cuda.synchronize()
print(f"z = [{N}, {C}, {H}, {W}] = {nbytes / 1e6:.1f} MB, {len(zs)} rotating copies")
