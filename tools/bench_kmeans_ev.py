"""Developer micro-benchmark: the k-means chain ALONE, hipEvent-timed (tools/bench_kmeans.py takes the wall clock around
label prep + replication + the chain + proxy construction + table copies, over 10 host-driven runs).  Here the label prep, the replicated
lists and the initial rows are produced once, outside every bracket; each repeat enqueues `n` chains back to back, every chain between its
own pair of events on the launching stream (aoc_kmeans_segmented_rep = the 20 Lloyd iterations; aoc_build_proxies bracketed separately), and
the per-chain durations are reported as min / median / max per repeat: a bimodal chain shows up as a spread, a slow first chain as a max.
Usage: python tools/bench_kmeans_ev.py <R> <F> [repeats=5] [chains per repeat=10] [cfg=cfg2]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aoc_amd  # noqa: E402
from aoc_amd import hotpath, ops, synthetic as syn  # noqa: E402
from aoc_amd.matching import KMEANS_ITERS  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
F = int(sys.argv[2]) if len(sys.argv) > 2 else 1
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
N = int(sys.argv[4]) if len(sys.argv) > 4 else 10
CFG = sys.argv[5] if len(sys.argv) > 5 else "cfg2"
cfg = syn.CONFIGS[CFG]
clip = syn.make_clip(cfg, 0, frames=(R - 1) * 5 + 1)
O, C = cfg.n_obj, cfg.c
emb = torch.from_numpy(clip["emb"][0::5][:R].copy()).cuda()
lab_ids = clip["lab"][0::5][:R]
lab = torch.from_numpy(np.stack([syn.one_hot(l, O) for l in lab_ids])).cuda()
mc = hotpath.MatchingConfig(CLUSTER_NUM=cfg.k, CLUSTER_LEVELS=[8, 16, 32] if CFG == "cfg3" else None)       # bench.py's CONFIG_LEVELS
levels = mc.cluster_levels
L, kmax = len(levels), max(levels)
counts = [int((lab_ids == o).sum()) for o in range(O)]
hw = emb.shape[1] * emb.shape[2]
pool = emb.reshape(R * hw, C)
prep = ops.label_prep(lab.reshape(R * hw, O))
cap = prep.obj_rows.numel()
rows_f, off_f, k_f = ops.kmeans_replicate_levels(prep.obj_rows, prep.obj_offsets, O, F * L, levels, rows_capacity=cap)
init = np.zeros((F * L * O, kmax), np.int32)
for f in range(F):
    for l, k in enumerate(levels):
        rows = syn.kmeans_init_rows(100 + f * 7 + l, counts, k)
        for o, r in enumerate(rows):
            if r is not None:
                init[(f * L + l) * O + o, :len(r)] = r
init = torch.from_numpy(init).cuda()
torch.cuda.synchronize()


def chain():
    cen, labels, _ = ops.kmeans_segmented(pool, rows_f, off_f, k_f, init, kmax, KMEANS_ITERS, rows_capacity=F * L * cap, n_rep=F * L)
    return cen, labels


def proxies(cen, labels):
    return ops.build_proxies(pool, prep.fg_rows, off_f, k_f, labels, cen)


for _ in range(3):
    proxies(*chain())
torch.cuda.synchronize()
print(f"k-means chain alone, {CFG} R={R} F={F} levels={levels}: {sum(counts)} rows x {F * L} replicas, {KMEANS_ITERS} iterations, {REPS} repeats x {N} chains", flush=True)
all_chain, all_prox = [], []
for rep in range(REPS):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(N)]
    t0 = time.perf_counter()
    for i in range(N):
        ev[i][0].record()
        cl = chain()
        ev[i][1].record()
        proxies(*cl)
        ev[i][2].record()
    t_host = (time.perf_counter() - t0) / N * 1e3
    torch.cuda.synchronize()
    t_wall = (time.perf_counter() - t0) / N * 1e3
    ch = sorted(e[0].elapsed_time(e[1]) for e in ev)
    px = sorted(e[1].elapsed_time(e[2]) for e in ev)
    all_chain += ch
    all_prox += px
    print(f"  repeat {rep}: chain min {ch[0]:.3f} median {ch[len(ch) // 2]:.3f} max {ch[-1]:.3f} ms | proxies median {px[len(px) // 2]:.3f} ms | "
          f"host enqueue {t_host:.3f} ms, wall {t_wall:.3f} ms per chain + proxies", flush=True)
    time.sleep(0.05)
all_chain.sort()
all_prox.sort()
print(f"k-means chain {CFG} R={R} F={F}: median {all_chain[len(all_chain) // 2]:.3f} ms (min {all_chain[0]:.3f}, max {all_chain[-1]:.3f}) per chain, "
      f"proxy construction median {all_prox[len(all_prox) // 2]:.3f} ms", flush=True)
# a library built with -DAOC_KS_STATS (tools/build_variant.sh kstat labels_kmeans.hip "-DAOC_KS_STATS -DAOC_DEV"; AOC_LIB_FILE=libaoc_hip_kstat.so) counts what
# the stitch did with the tail chunks' summaries over everything above
try:
    import ctypes
    fn = aoc_amd._lib.lib().aoc_debug_ks_stats
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_ulonglong * 8)()
    fn(buf, 1)
    v, ok, cold = int(buf[5]), int(buf[6]), int(buf[7])
    print(f"  stitch: {v} (tail chunk, feature) visits, {ok / max(v, 1):.4f} verified summaries, {(v - ok - cold) / max(v, 1):.4f} refolded with prefetched rows, "
          f"{cold / max(v, 1):.4f} refolded without a prefetch (mispredicted)", flush=True)
except AttributeError:
    pass
