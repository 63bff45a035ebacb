"""Developer micro-benchmark: the 20-iteration k-means chain alone (aoc_kmeans_segmented via hotpath.launch_cluster_proxies_batch) at cfg2 size.
Usage: python tools/bench_kmeans.py <pool frames R> <frames per chain F>"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import aoc_amd
from aoc_amd import hotpath, synthetic as syn

R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
F = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = syn.CONFIGS["cfg2"]
clip = syn.make_clip(cfg, 0, frames=(R - 1) * 5 + 1)
O, C = cfg.n_obj, cfg.c
emb = torch.from_numpy(clip["emb"][0::5][:R].copy()).cuda()
lab_ids = clip["lab"][0::5][:R]
lab = torch.from_numpy(np.stack([syn.one_hot(l, O) for l in lab_ids])).cuda()
mc = hotpath.MatchingConfig()
counts = [int((lab_ids == o).sum()) for o in range(O)]
inits = []
for f in range(F):
    rows = syn.kmeans_init_rows(100 + f, counts, mc.CLUSTER_NUM)
    init = np.zeros((O, mc.CLUSTER_NUM), np.int32)
    for o, r in enumerate(rows):
        if r is not None:
            init[o, :len(r)] = r
    inits.append(torch.from_numpy(init).cuda())
import ctypes, os
NCU = int(os.environ.get("KM_CUS", "0"))                   # run the chain on a stream masked to the first KM_CUS CUs (what it finds free next to a dense kernel)
if NCU > 0:
    hip = ctypes.CDLL("libamdhip64.so")
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (n_cu + 31) // 32
    mask = [0] * words
    for cu in range(NCU):
        mask[cu // 32] |= 1 << (cu % 32)
    handle = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), ctypes.c_uint32(words), (ctypes.c_uint32 * words)(*mask)) == 0
    side = torch.cuda.ExternalStream(handle.value)
else:
    side = torch.cuda.Stream()
def run():
    if F == 1:
        return hotpath.launch_cluster_proxies(mc, emb, lab, inits[0], side)
    return hotpath.launch_cluster_proxies_batch(mc, emb, lab, inits, side)
for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    run()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
rows = sum(counts)
print(f"k-means chain R={R} F={F} CUs={NCU or 'all'}: {ms:.3f} ms per chain ({ms / F:.3f} per frame), {rows} rows x {F} replicas", flush=True)

NS = int(os.environ.get("KM_STREAMS", "0"))                # KM_STREAMS=n: n chains at a time, each on its own stream (do concurrent chains overlap?)
if NS > 1:
    sides = [torch.cuda.Stream(priority=-1) for _ in range(NS)]
    def run_all():
        for sd in sides:
            hotpath.launch_cluster_proxies(mc, emb, lab, inits[0], sd)
    for _ in range(2):
        run_all()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run_all()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"  {NS} chains at a time on {NS} streams: {ms:.3f} ms per round = {ms / NS:.3f} ms per chain", flush=True)
