#!/usr/bin/env python3
"""Benchmark of the AOC-Net matching + calibration hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one frame of one synthetic video sequence per stream (--streams, default 1) through the
whole hot path: label prep, scipy-exact k-means proxies, proxy / dense / local matching, fg->bg,
the 24-channel proto-mask tensor, then the ten IA gates and four conditioning blocks at the
decoder's activation shapes.  Workload = BASELINE.json configs[1] (cfg2): 480p -> 121x213 stride-4
maps, 3 objects + background, K = 16 proxies, 60-frame clips whose reference pool grows by one frame
every MEM_EVERY = 5 frames (R = 1..12), exactly as the reference's eval loop does
(eval_manager_mm.py:309-361).  Inputs (feature maps, label maps, k-means initial rows, decoder
activations) are synthetic, seeded and resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement): whole-job frames/s, the roofline of
the dominant kernel measured with HIP events inside the timed region, and a CPU baseline (the oracle
timed on the host cores, rank 0, N = 1 only).
"""
import argparse
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # main + dense + k-means stream per sequence: more than the default 4 hardware queues
import ctypes
import json
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import aoc_amd  # noqa: E402
from aoc_amd import hotpath, ops, sharding  # noqa: E402
from aoc_amd import synthetic as syn  # noqa: E402

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_F16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_f16, dense
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense


class HipEventPairs:
    """hipEvent_t pairs created through the HIP runtime (ctypes) for aoc_dense_match_set_probe: the library records them
    immediately around its matrix kernel, on the stream of the call."""

    def __init__(self):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self.pairs = []

    def arm(self):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(a)) == 0 and self.hip.hipEventCreate(ctypes.byref(b)) == 0
        aoc_amd._lib.check(aoc_amd._lib.lib().aoc_dense_match_set_probe(a, b), "aoc_dense_match_set_probe")
        self.pairs.append((a, b))

    def elapsed_ms(self):
        out = []
        for a, b in self.pairs:
            ms = ctypes.c_float()
            if self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0:
                out.append(ms.value)
            self.hip.hipEventDestroy(a)
            self.hip.hipEventDestroy(b)
        self.pairs = []
        return out


class OpTimer:
    """HIP-event timing of every call of selected aoc_amd.ops functions on the stream they are
    launched on (torch's current stream), inside the timed region."""

    def __init__(self, names):
        self.names = names
        self.records = {n: [] for n in names}
        self.meta = {n: [] for n in names}
        self.enabled = False
        self._orig = {}
        self.kernel_probe = HipEventPairs()
        self.dense_done = None
        self.serialize_dense = True
        self.probed = ("dense_match_min_split", "dense_match_min")

    def install(self, meta_fns):
        for n in self.names:
            fn = getattr(ops, n)
            self._orig[n] = fn

            def wrapped(*a, _fn=fn, _n=n, **k):
                if not self.enabled:
                    return _fn(*a, **k)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if _n in self.probed and self.dense_done is not None and self.serialize_dense:
                    # only one dense kernel fits per CU, so the dense kernels of the two sequences run back to back anyway;
                    # making that order explicit keeps the queueing of one behind the other out of the timed interval
                    torch.cuda.current_stream().wait_event(self.dense_done)
                e0.record()
                if _n in self.probed:
                    self.kernel_probe.arm()
                out = _fn(*a, **k)
                e1.record()
                if _n in self.probed:
                    self.dense_done = e1
                self.records[_n].append((e0, e1))
                self.meta[_n].append(meta_fns[_n](*a, **k) if _n in meta_fns else None)
                return out

            setattr(ops, n, wrapped)

    def summary(self):
        out = {}
        for n in self.names:
            ms = [a.elapsed_time(b) for a, b in self.records[n]]
            if ms:
                out[n] = dict(calls=len(ms), total_ms=float(np.sum(ms)), avg_ms=float(np.mean(ms)), meta=self.meta[n])
        return out


class ClipWorkload:
    """One synthetic sequence of a config, resident on the GPU, stepped with the reference's memory policy."""

    def __init__(self, cfg, seed, device, mc, overlap=True):
        self.cfg, self.mc, self.dev = cfg, mc, device
        # the k-means branch (a long chain of small launches) runs on a high-priority side stream,
        # concurrently with the MFMA-bound dense matching on the main stream
        self.side = torch.cuda.Stream(device=device, priority=-1) if overlap else None
        clip = syn.make_clip(cfg, seed)
        O = cfg.n_obj
        self.emb = torch.from_numpy(clip["emb"]).to(device)                                   # [T,h,w,C]
        self.lab_ids = clip["lab"]
        self.lab = torch.from_numpy(np.stack([syn.one_hot(l, O) for l in clip["lab"]])).to(device)   # [T,h,w,O]
        self.T = cfg.frames
        self.bias = torch.zeros(O, device=device)
        rmax = 1 + (self.T - 1) // mc.MEM_EVERY + 1
        self.pool_emb = torch.empty(rmax, cfg.h, cfg.w, cfg.c, device=device)
        self.pool_lab = torch.empty(rmax, cfg.h, cfg.w, O, device=device)
        # per-frame k-means initial rows (inputs): permutation(n_i)[:K_i] with the sticky K rule
        self.init_rows = {}
        ref_idx = [0]
        for t in range(1, self.T):
            counts = [int(sum((self.lab_ids[i] == o).sum() for i in ref_idx)) for o in range(O)]
            rows = syn.kmeans_init_rows(seed * 100003 + t, counts, mc.CLUSTER_NUM)
            init = np.zeros((O, mc.CLUSTER_NUM), np.int32)
            for o, r in enumerate(rows):
                if r is not None:
                    init[o, :len(r)] = r
            self.init_rows[t] = (torch.from_numpy(init).to(device), rows)
            if t % mc.MEM_EVERY == 0:
                ref_idx.append(t)
        self.dense_state = {"capacity_frames": rmax}      # fp16 split records of the pool, converted once per appended frame
        self.ahead = {}                                    # frame -> adaptive proxies already enqueued on a side stream
        self.pool_event = None                             # recorded after the last change of the pool
        self.chains = mc.MEM_EVERY                         # k-means chains enqueued ahead (bench --chains)
        self.dense_stream = None                           # CU-masked stream for the dense kernel alone
        self.reuse_proxies = False                         # non-parity mode: one k-means per pool state
        self.cached_ahead = None
        self.reset()

    def reset(self):
        self.t, self.R = 1, 1
        self.dense_state["frames"] = 0
        self.dense_state.pop("ref_pool", None)
        self.cached_ahead = None
        self.pool_emb[0].copy_(self.emb[0])
        self.pool_lab[0].copy_(self.lab[0])
        self.pool_event = torch.cuda.Event()
        self.pool_event.record()

    def refs(self):
        return self.pool_emb[:self.R], self.pool_lab[:self.R]

    def pretouch(self, gates, acts, dense_precision, pipeline):
        """Setup, not a benchmark step: one frame at EVERY pool size the clip will reach, largest first, so that torch's
        caching allocator already owns a block for every request of the timed region (otherwise each pool growth calls
        hipMalloc, which synchronises the device).  The pool is filled with copies of frame 0 for this; reset() restores it."""
        rmax = self.pool_emb.shape[0]
        self.pool_emb[:] = self.emb[0]
        self.pool_lab[:] = self.lab[0]
        for R in range(rmax, 0, -1):
            self.R, self.t = R, self.T - 1
            self.pool_event = torch.cuda.Event()
            self.pool_event.record()
            self.ahead.clear()
            self.dense_state["frames"] = 0
            self.dense_state.pop("ref_pool", None)
            ref_emb, ref_lab = self.refs()
            counts = [int((self.lab_ids[0] == o).sum()) * R for o in range(self.cfg.n_obj)]
            rows = syn.kmeans_init_rows(12345, counts, self.mc.CLUSTER_NUM)
            init = np.zeros((self.cfg.n_obj, self.mc.CLUSTER_NUM), np.int32)
            for o, r in enumerate(rows):
                if r is not None:
                    init[o, :len(r)] = r
            init = torch.from_numpy(init).to(self.dev)
            if self.side is not None and pipeline:
                ahead = hotpath.launch_cluster_proxies(self.mc, ref_emb, ref_lab, init, self.side, wait_event=self.pool_event)
                for nb in range(2, self.chains):               # batched chains of every size the run can ask for
                    hotpath.launch_cluster_proxies_batch(self.mc, ref_emb, ref_lab, [init] * nb, self.side, wait_event=self.pool_event)
                feat, head, _ = hotpath.proto_mask_features(self.mc, ref_emb, ref_lab, self.emb[1], self.lab[1], self.emb[2], self.bias,
                                                            cluster_ahead=ahead, dense_state=self.dense_state, dense_precision=dense_precision,
                                                            dense_stream=self.dense_stream)
            else:
                feat, head, _ = hotpath.proto_mask_features(self.mc, ref_emb, ref_lab, self.emb[1], self.lab[1], self.emb[2], self.bias,
                                                            cluster_state=dict(init_rows=init), side_stream=self.side,
                                                            dense_state=self.dense_state, dense_precision=dense_precision)
            gates(acts, head)
            torch.cuda.synchronize()
        self.ahead.clear()
        self.reset()

    def advance(self):
        """eval_manager_mm.py:309-312,356-361: append the frame to the pool every MEM_EVERY frames."""
        if self.t % self.mc.MEM_EVERY == 0:
            self.pool_emb[self.R].copy_(self.emb[self.t])
            self.pool_lab[self.R].copy_(self.lab[self.t])
            self.R += 1
            self.pool_event = torch.cuda.Event()
            self.pool_event.record()                       # the next k-means chain must see the appended frame
        self.t += 1
        if self.t >= self.T:
            self.reset()


def make_activations(gates, O, h, w, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return [torch.randn(O, c, hh, ww, generator=g).to(device) for (_, c, hh, ww, _) in gates.plan(h, w)]


def frame_step(wl, gates, acts, dense_precision="split", pipeline=True):
    ref_emb, ref_lab = wl.refs()
    t = wl.t
    if wl.side is not None and pipeline:
        # the k-means chain of a frame only depends on the pool (which changes every MEM_EVERY frames): the chains of all
        # frames that will see the same pool are enqueued on side streams right after the pool update and run under the
        # other work of the frames before them
        if wl.reuse_proxies and wl.cached_ahead is not None and wl.cached_ahead.R == wl.R:
            # NON-PARITY mode: the adaptive proxies computed for this pool are reused until the pool changes,
            # instead of re-clustering the unchanged pool with fresh initial rows for every frame like the reference
            ahead = wl.cached_ahead
        else:
            if t not in wl.ahead:
                launch_chains(wl)
            ahead = wl.ahead.pop(t)
            wl.cached_ahead = ahead if wl.reuse_proxies else None
        if not wl.reuse_proxies and t % wl.mc.MEM_EVERY != 0 and t + 1 < wl.T and (t + 1) not in wl.ahead:
            # the next frame sees the same pool: its chain goes onto the side stream now, behind this frame's chain, and
            # does not wait for anything this frame still has to do on the main stream
            wl.ahead[t + 1] = hotpath.launch_cluster_proxies(wl.mc, ref_emb, ref_lab, wl.init_rows[t + 1][0], wl.side, wait_event=wl.pool_event)
        feat, head, _ = hotpath.proto_mask_features(wl.mc, ref_emb, ref_lab, wl.emb[t - 1], wl.lab[t - 1], wl.emb[t], wl.bias,
                                                    cluster_ahead=ahead, dense_state=wl.dense_state, dense_precision=dense_precision,
                                                    dense_stream=wl.dense_stream)
    else:
        feat, head, _ = hotpath.proto_mask_features(wl.mc, ref_emb, ref_lab, wl.emb[t - 1], wl.lab[t - 1], wl.emb[t], wl.bias,
                                                    cluster_state=dict(init_rows=wl.init_rows[t][0]), side_stream=wl.side,
                                                    dense_state=wl.dense_state, dense_precision=dense_precision)
    outs = gates(acts, head)
    wl.advance()                                           # pool append / sequence restart happen here (after the frame's outputs)
    if wl.side is not None and pipeline and wl.t not in wl.ahead and not (wl.reuse_proxies and wl.cached_ahead is not None and wl.cached_ahead.R == wl.R):
        launch_chains(wl)
    return feat, outs


def launch_chains(wl):
    """Enqueue the k-means chains of frame wl.t and of the following frames up to the next pool update (they all see the
    pool as it is now) on the side stream: the first frame alone (it is needed first), the others batched into one chain."""
    ref_emb, ref_lab = wl.refs()
    t = wl.t
    for k in [k for k in wl.ahead if k < t or k >= wl.T]:
        del wl.ahead[k]
    frames = [t]
    while frames[-1] % wl.mc.MEM_EVERY != 0 and frames[-1] + 1 < wl.T and len(frames) < wl.chains:
        frames.append(frames[-1] + 1)                      # a frame with t % MEM_EVERY == 0 is appended: later ones see another pool
    wl.ahead[t] = hotpath.launch_cluster_proxies(wl.mc, ref_emb, ref_lab, wl.init_rows[t][0], wl.side, wait_event=wl.pool_event)
    if len(frames) > 1:
        rest = hotpath.launch_cluster_proxies_batch(wl.mc, ref_emb, ref_lab, [wl.init_rows[f][0] for f in frames[1:]], wl.side,
                                                    wait_event=wl.pool_event)
        for f, a in zip(frames[1:], rest):
            wl.ahead[f] = a


def _block_weights(mod):
    sd = {k: v.detach().cpu() for k, v in mod.state_dict().items()}
    return {"CL_1.phi_w": sd["CL_1.phi_layer.weight"].reshape(-1), "CL_1.phi_b": sd["CL_1.phi_layer.bias"],
            "CL_1.mlp_w": sd["CL_1.mlp_layer.weight"], "CL_1.mlp_b": sd["CL_1.mlp_layer.bias"],
            "CL_2.mlp_w": sd["CL_2.mlp_layer.weight"], "CL_2.mlp_b": sd["CL_2.mlp_layer.bias"],
            "CL_3.mlp_w": sd["CL_3.mlp_layer.weight"], "CL_3.mlp_b": sd["CL_3.mlp_layer.bias"],
            "mlp_w": sd["mlp_layer.weight"], "mlp_b": sd["mlp_layer.bias"]}


DENSE_SUBSAMPLE = 16


def cpu_baseline(cfg, mc, seed, gates, acts_cpu):
    """The oracle (a port of the reference path, proved equal to it on the golden vectors) timed on the
    host cores for ONE frame of the same workload at R = 1 (the cheapest frame of the clip), bounded to
    a few tens of seconds: the dense branch (linear in query pixels) runs on every 16th query pixel and
    is scaled by 16; each distinct calibration gate shape is timed once and multiplied by its count.
    Returns the per-branch features (for a parity spot check) and the timings."""
    from oracle import calibration as ocal
    from oracle import matching as om
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    clip = syn.make_clip(cfg, seed, frames=2)
    O = cfg.n_obj
    emb = torch.from_numpy(clip["emb"])
    lab = torch.from_numpy(np.stack([syn.one_hot(l, O) for l in clip["lab"]]))
    counts = [int((clip["lab"][0] == o).sum()) for o in range(O)]
    rows = syn.kmeans_init_rows(seed * 100003 + 1, counts, mc.CLUSTER_NUM)
    bias = torch.zeros(O)
    mld = list(mc.MODEL_MULTI_LOCAL_DISTANCE)
    tm, feats = {}, {}

    def timed(key, fn):
        t0 = time.perf_counter()
        out = fn()
        tm[key] = time.perf_counter() - t0
        return out

    ref_flat, lab_flat = emb[0].reshape(-1, cfg.c), lab[0].reshape(-1, O)
    q_sub = emb[1].reshape(-1, cfg.c)[::DENSE_SUBSAMPLE]
    feats["dense_sub"] = timed("dense", lambda: om.proto_transform(om.nearest_neighbor_features_per_object(ref_flat, q_sub, lab_flat).squeeze(-1), bias.view(1, -1)))
    tm["dense"] *= DENSE_SUBSAMPLE
    feats["cluster"] = timed("cluster", lambda: om.global_matching_for_eval_cluster([emb[0]], emb[1], [lab[0]], 4, bias, init_rows=rows))
    feats["local"] = timed("local", lambda: om.local_matching(emb[0], emb[1], lab[0], bias, mld))
    ref_e, ref_l = [emb[0].permute(2, 0, 1).unsqueeze(0)], [lab[0].permute(2, 0, 1).unsqueeze(1)]
    head, ref_pos, _, prev_pos, _ = timed("pool", lambda: ocal.attention_head_for_eval_p_m(
        ref_e, ref_l, emb[0].permute(2, 0, 1).unsqueeze(0).expand(O, -1, -1, -1), ref_l[0], mc.MODEL_EPSILON))
    feats["proxy"] = timed("proxy", lambda: om.global_matching_for_eval_proxy(ref_pos, emb[1], [lab[0]], 4, bias))
    feats["local_proxy"] = timed("local_proxy", lambda: om.local_matching(torch.matmul(lab[0], prev_pos), emb[1], lab[0], bias, mld))
    t_match = sum(tm.values())
    t_cal, seen = 0.0, {}
    for (name, c, hh, ww, extra), x in zip(gates.plan(cfg.h, cfg.w), acts_cpu):
        mod = getattr(gates, name)
        key = (name.startswith("CLB"), c, hh, ww, extra)
        if key not in seen:
            t0 = time.perf_counter()
            if name.startswith("CLB"):
                ocal.conditioning_block(x, head, _block_weights(mod), mc.BETA_PERCENTAGE)
            else:
                hd = head
                if extra:
                    px = x.mean(dim=(2, 3))
                    hd = torch.cat([head, px.sum(0, keepdim=True) - px], 1)
                sd = {k: v.detach().cpu() for k, v in mod.state_dict().items()}
                ocal.ia_gate(x, hd, sd["IA.weight"], sd["IA.bias"])
            seen[key] = time.perf_counter() - t0
        t_cal += seen[key]
    return feats, rows, tm, t_match, t_cal, threads


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=59)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=list(syn.CONFIGS))
    ap.add_argument("--streams", type=int, default=2, help="independent sequences stepped concurrently on separate HIP streams")
    ap.add_argument("--dense", default="split", choices=["split", "fp32"],
                    help="dense-matching arithmetic: fp16-split products with fp32 accumulate (fp32-equivalent) or exact-fp32 MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exact-run", action="store_true",
                    help="add an informational second region: the same K steps with the exact-fp32 dense kernel (reported as exact_fp32_dense_run)")
    ap.add_argument("--cu-reserve", type=int, default=64,
                    help="keep the main streams off this many CUs (HIP CU mask) so the side-stream k-means chain always finds free CUs")
    ap.add_argument("--chains", type=int, default=3,
                    help="frames whose k-means is enqueued right after a pool update (1 = only the next frame; the others are batched into one chain)")
    ap.add_argument("--dense-stream", dest="mask_main", action="store_false",
                    help="put only the dense kernel under the CU mask (on its own stream) instead of the whole main stream "
                         "(measured slower: the unmasked light kernels then take the reserved CUs from the k-means chains)")
    ap.add_argument("--reuse-proxies", action="store_true",
                    help="NON-PARITY mode (SURVEY 8f-3): cluster the pool once per pool update instead of once per frame; the JSON "
                         "line then says so in config.proxy_mode and is not comparable with the default")
    ap.add_argument("--no-dense-order", action="store_true",
                    help="do not order the sequences' dense kernels explicitly (their live timing then includes queueing behind each other)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="start a frame's k-means chain with the frame instead of right after the previous frame's pool update")
    ap.add_argument("--no-overlap", action="store_true", help="run the k-means branch on the main stream (no intra-frame stream overlap)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL over xGMI
    if args.cu_reserve > 0:
        os.environ.setdefault("AOC_DENSE_CUS", str(256 - args.cu_reserve))   # the dense kernel sizes its grid in whole rounds of CUs
    aoc_amd._lib.lib()

    cfg = syn.CONFIGS[args.config]
    mc = hotpath.MatchingConfig(CLUSTER_NUM=cfg.k)
    torch.manual_seed(0)
    gates = hotpath.CalibrationGates(mc).to(dev)
    n_streams = max(1, args.streams)
    # sequences are sharded over ranks: rank r owns sequences r*n_streams .. (+n_streams)
    workloads = [ClipWorkload(cfg, seed=1 + rank * n_streams + s, device=dev, mc=mc, overlap=not args.no_overlap) for s in range(n_streams)]
    for wl in workloads:
        wl.chains = 1 if args.reuse_proxies else max(1, min(args.chains, mc.MEM_EVERY))
        wl.reuse_proxies = args.reuse_proxies
    acts = make_activations(gates, cfg.n_obj, cfg.h, cfg.w, dev, seed=7)
    def make_main_stream():
        """Main stream of one sequence.  With --cu-reserve N its workgroups are kept off N of the 256 CUs (HIP CU mask), so
        that the latency-bound k-means chain on the side stream always finds free CUs next to the dense kernel, whose
        blocks fill a CU's register file and do not yield."""
        if args.cu_reserve <= 0:
            return torch.cuda.Stream(device=dev)
        hip = ctypes.CDLL("libamdhip64.so")
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        words = (n_cu + 31) // 32
        mask = [0] * words
        for cu in range(n_cu - args.cu_reserve):           # contiguous: interleaved masks were observed to be ignored
            mask[cu // 32] |= 1 << (cu % 32)
        arr = (ctypes.c_uint32 * words)(*mask)
        handle = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), ctypes.c_uint32(words), arr)
        if rc != 0 or not handle.value:
            print(f"bench: hipExtStreamCreateWithCUMask failed ({rc}); running without the CU reservation", file=sys.stderr)
            args.cu_reserve = 0
            os.environ.pop("AOC_DENSE_CUS", None)
            return torch.cuda.Stream(device=dev)
        return torch.cuda.ExternalStream(handle.value, device=dev)

    if args.mask_main:
        streams = [make_main_stream() for _ in range(n_streams)] if (n_streams > 1 or args.cu_reserve > 0) else [torch.cuda.current_stream()]
        dense_streams = [None] * n_streams
    else:
        # only the dense kernel runs under the CU mask (its own stream per sequence); everything else may use every CU
        streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 else [torch.cuda.current_stream()]
        dense_streams = [make_main_stream() if args.cu_reserve > 0 else None for _ in range(n_streams)]
    for wl, ds in zip(workloads, dense_streams):
        wl.dense_stream = ds

    hw, C, O = cfg.h * cfg.w, cfg.c, cfg.n_obj

    def meta_dense(query_flat, pool, prep, *a, **k):
        m, n = query_flat.shape[0], pool.shape[0]      # synthetic labels: every pool pixel is kept
        return dict(flops=2.0 * m * n * C, bytes=(m + n) * C * 4 + n * 4 + m * O * 4)

    def meta_dense_split(query_flat, query_split, pool, pool_split, prep, *a, **k):
        return meta_dense(query_flat, pool[:prep.n], prep)

    def meta_proxy(query_flat, proxies, *a, **k):
        m, npx = query_flat.shape[0], proxies.shape[0]
        return dict(flops=2.0 * m * npx * C, bytes=m * C * 4 + npx * C * 4 + m * (3 * O) * 4)

    def meta_kmeans(pool, rows, seg_offsets, seg_k, init_rows, kmax, iters=20, rows_capacity=None):
        n = pool.shape[0]
        return dict(flops=2.0 * iters * n * kmax * C, bytes=float(iters) * n * C * 4 * 2 + n * 4)

    # HIP-event pairs only around the ops the roofline objects need (an event pair costs ~25 us of host time, and the host
    # enqueues ~60 ops per frame); every kernel's duration is in the rocprofv3 summary under profiles/
    timer = OpTimer(["dense_match_min", "dense_match_min_split", "proxy_corr_min", "kmeans_segmented", "local_window_match"])
    timer.serialize_dense = not args.no_dense_order
    timer.install(dict(dense_match_min=meta_dense, dense_match_min_split=meta_dense_split, proxy_corr_min=meta_proxy, kmeans_segmented=meta_kmeans))

    def run_steps(n):
        for _ in range(n):
            for wl, st in zip(workloads, streams):
                if n_streams > 1 or (args.cu_reserve > 0 and args.mask_main):
                    with torch.cuda.stream(st):
                        frame_step(wl, gates, acts, args.dense, not args.no_pipeline)
                else:
                    frame_step(wl, gates, acts, args.dense, not args.no_pipeline)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()

    with torch.no_grad():
        for wl, st in zip(workloads, streams):             # allocator pre-touch at the largest pool size (setup)
            with torch.cuda.stream(st):
                wl.pretouch(gates, acts, args.dense, not args.no_pipeline)
        run_steps(args.warmup)
        barrier()
        timer.enabled = True                 # HIP events around every op, on the stream the op is launched on
        t0 = time.perf_counter()
        if os.environ.get("AOC_BENCH_PROFILE"):
            import cProfile, pstats
            pr = cProfile.Profile()
            pr.enable()
            run_steps(args.steps)
            pr.disable()
            host_s = time.perf_counter() - t0
            pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
            print(f"host time to enqueue {args.steps} steps: {host_s * 1e3:.1f} ms", file=sys.stderr)
        else:
            run_steps(args.steps)
        host_enqueue_s = time.perf_counter() - t0
        barrier()
        elapsed = time.perf_counter() - t0
        timer.enabled = False

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed_max = float(el.item())
    frames_local = args.steps * n_streams
    metrics = sharding.allreduce_metrics(dict(frames=frames_local, objects=frames_local * (O - 1), gpu_seconds=elapsed), device=dev)

    probe_ms = timer.kernel_probe.elapsed_ms()

    # second, informational region (N = 1 only): the same K steps with the exact-fp32 dense kernel (`--dense fp32`), so that the
    # line also carries the figure of the all-fp32 arithmetic next to the headline
    exact = None
    if world == 1 and args.dense == "split" and args.exact_run:
        with torch.no_grad():
            for wl in workloads:
                wl.reset()
                wl.ahead.clear()
            saved = args.dense
            args.dense = "fp32"
            run_steps(min(args.warmup, 2))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(args.steps)
            torch.cuda.synchronize()
            e2 = time.perf_counter() - t1
            args.dense = saved
        exact = dict(value=round(args.steps * n_streams / e2, 3), unit="frames/s", ms_per_step=round(e2 / args.steps * 1e3, 4),
                     note="same workload and steps with aoc_dense_match_min (v_mfma_f32_16x16x4_f32) instead of the fp16-split kernel")
    if rank == 0:
        summ = timer.summary()
        kernels = {}
        for name, s in summ.items():
            k = dict(calls=s["calls"], avg_ms=round(s["avg_ms"], 4), total_ms=round(s["total_ms"], 3))
            metas = [m for m in s["meta"] if m]
            if metas:
                fl = float(np.mean([m["flops"] for m in metas]))
                by = float(np.mean([m["bytes"] for m in metas]))
                k.update(avg_flops=fl, avg_bytes=by, tflops=round(fl / (s["avg_ms"] * 1e-3) / 1e12, 3),
                         gbs=round(by / (s["avg_ms"] * 1e-3) / 1e9, 2))
            kernels[name] = k
        roofline = None
        dense_ops = [n for n in ("dense_match_min_split", "dense_match_min") if n in kernels]
        if dense_ops:
            # the single kernel with the largest share of GPU time (profiles/: dense_split_kernel); the k-means op is a chain of
            # ~160 small launches per call and is reported as its own object below
            dom = max(dense_ops, key=lambda n: kernels[n]["total_ms"])
            k = dict(kernels[dom])
            if probe_ms:
                # the matrix kernel alone (hipEvents recorded by the library right around its launch); k["avg_ms"] spans the op
                k["op_avg_ms"] = k["avg_ms"]
                k["avg_ms"] = round(float(np.mean(probe_ms)), 4)
                k["tflops"] = round(k["avg_flops"] / (k["avg_ms"] * 1e-3) / 1e12, 3)
            if dom == "dense_match_min":
                roofline = dict(kernel="dense_match_partial_kernel (aoc_dense_match_min)", bound="mfma", achieved=k["tflops"],
                                peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s", frac=round(k["tflops"] / PEAK_FP32_MFMA_TFLOPS, 4), traffic=None,
                                avg_launch_ms=k["avg_ms"], algorithmic_flops_per_launch=k["avg_flops"])
            else:
                # algorithmic flops (2 m n C) against the fp16 pipe the kernel runs on; the instruction stream
                # executes 3 split products on K padded 100 -> 112, i.e. 3.36x the algorithmic flops
                executed = k["tflops"] * 3.0 * 112.0 / C
                roofline = dict(kernel="dense_split_kernel (aoc_dense_match_min_split)", bound="mfma", achieved=k["tflops"],
                                peak=PEAK_F16_MFMA_TFLOPS, unit="TFLOP/s", frac=round(k["tflops"] / PEAK_F16_MFMA_TFLOPS, 4), traffic=None,
                                avg_launch_ms=k["avg_ms"], algorithmic_flops_per_launch=k["avg_flops"],
                                executed_tflops=round(executed, 1), pipe_frac=round(executed / PEAK_F16_MFMA_TFLOPS, 4),
                                note="fp32-equivalent products from 3 fp16 MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulate); frac prices the "
                                     "ALGORITHMIC fp32 flops against the dense fp16 peak, pipe_frac the executed ones; avg_launch_ms = hipEvents "
                                     "recorded by the library immediately around the kernel while the other sequence's stream shares the GPU",
                                op_avg_ms=k.get("op_avg_ms"), traffic_source=None)
                if args.cu_reserve > 0:
                    # the kernel is launched on a stream whose CU mask leaves cu_reserve CUs to the k-means chains
                    cus = 256 - args.cu_reserve
                    roofline["cus_available_to_kernel"] = cus
                    roofline["pipe_frac_of_available_cus"] = round(executed / (PEAK_F16_MFMA_TFLOPS * cus / 256.0), 4)
                pmc_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_dense_split.json")
                if os.path.exists(pmc_file):
                    # PMC counters cannot be read from inside the run: separate rocprofv3 --pmc passes of this command, committed
                    with open(pmc_file) as fh:
                        pmc = json.load(fh)
                    roofline["traffic"] = pmc["traffic_bytes_per_launch"]
                    roofline["traffic_source"] = "profiles/r01_pmc_dense_split.json (FETCH_SIZE x2 + WRITE_SIZE per dispatch, separate --pmc passes)"
        km = kernels.get("kmeans_segmented")
        km_roof = None
        if km and "gbs" in km:
            km_roof = dict(kernel="aoc_kmeans_segmented_ex (20 Lloyd iterations: assign+rank, block scan, scatter, ordered sums)", bound="hbm",
                           achieved=km["gbs"], peak=PEAK_HBM_GBS, unit="GB/s", frac=round(km["gbs"] / PEAK_HBM_GBS, 4),
                           avg_launch_ms=km["avg_ms"], algorithmic_bytes_per_launch=km["avg_bytes"],
                           note="a dependent chain of ~160 launches whose ordered float32 sums are latency-bound by construction")
        corr = kernels.get("proxy_corr_min")
        corr_roof = None
        if corr and "gbs" in corr:
            corr_roof = dict(kernel="proxy_corr_min_kernel", bound="hbm", achieved=corr["gbs"], peak=PEAK_HBM_GBS, unit="GB/s",
                             frac=round(corr["gbs"] / PEAK_HBM_GBS, 4), avg_launch_ms=corr["avg_ms"], algorithmic_bytes_per_launch=corr["avg_bytes"],
                             note="in-run figure: the event pair also spans the time the launch waits behind the other streams' kernels")
            # the kernel by itself: the same launch (one frame: 132 proxies x 25 773 pixels) 50 times back to back on an idle GPU
            wl = workloads[0]
            with torch.no_grad():
                torch.cuda.synchronize()
                kmax = mc.CLUSTER_NUM
                table = torch.randn(O * 2 * kmax + O, C, device=dev) * 0.3
                sqn = table.pow(2).sum(1)
                feat = torch.empty(O, mc.proto_channels, cfg.h, cfg.w, device=dev)
                stride = mc.proto_channels * hw
                sb = [(o * 2 + f) * kmax for o in range(O) for f in range(2)] + [O * 2 * kmax + o for o in range(O)]
                ss = [kmax] * (2 * O) + [1] * O
                so = [o * stride + (1 + f) * hw for o in range(O) for f in range(2)] + [o * stride + 3 * hw for o in range(O)]
                bias3 = torch.zeros(3 * O, device=dev)
                q = wl.emb[1].reshape(hw, C)
                run = lambda: timer._orig["proxy_corr_min"](q, table, sqn, sb, ss, so, bias3, feat, 1, True)
                for _ in range(5):
                    run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    run()
                e1.record()
                torch.cuda.synchronize()
                iso_ms = e0.elapsed_time(e1) / 50
            iso_gbs = corr["avg_bytes"] / (iso_ms * 1e-3) / 1e9
            corr_roof.update(isolated_avg_launch_ms=round(iso_ms, 4), isolated_achieved=round(iso_gbs, 1), isolated_frac=round(iso_gbs / PEAK_HBM_GBS, 4))

        cpu = None
        parity = None
        if world == 1 and not args.no_cpu_baseline:
            acts_cpu = [a.cpu() for a in acts]
            feats, rows, tm, t_match, t_cal, threads = cpu_baseline(cfg, mc, 1, gates, acts_cpu)
            br = ", ".join(f"{k} {v:.2f}" for k, v in tm.items())
            cpu = dict(value=round(1.0 / (t_match + t_cal), 5), unit="frames/s", cores=threads, kind="port",
                       sample=f"1 frame of {cfg.name} at R=1 (the cheapest frame; the GPU figure averages R=1..12): matching {t_match:.2f} s "
                              f"[{br}; dense timed on every {DENSE_SUBSAMPLE}th query pixel and scaled x{DENSE_SUBSAMPLE}] + calibration gates "
                              f"{t_cal:.2f} s (each distinct gate shape timed once x its count); torch CPU fp32 + C k-means oracle, {threads} threads")
            # parity spot check of the same frame on the GPU: every branch against the oracle, and a surrogate
            # mask (argmin over objects of the dense-matching channel) on the sub-sampled pixels
            wl = workloads[0]
            with torch.no_grad():
                fg, _, _ = hotpath.proto_mask_features(mc, wl.emb[:1], wl.lab[:1], wl.emb[0], wl.lab[0], wl.emb[1], wl.bias, init_rows=rows)
            fg = fg.cpu()
            chs = hotpath.channel_slices(mc)
            nl = len(mc.MODEL_MULTI_LOCAL_DISTANCE)
            diffs = {
                "dense": float((fg[:, 0].reshape(O, -1)[:, ::DENSE_SUBSAMPLE].t() - feats["dense_sub"]).abs().max()),
                "cluster": float((fg[:, chs["cluster"]:chs["cluster"] + 2] - feats["cluster"][0].permute(2, 3, 0, 1)).abs().max()),
                "proxy": float((fg[:, chs["proxy"]:chs["proxy"] + 1] - feats["proxy"][0].permute(2, 3, 0, 1)).abs().max()),
                "local": float((fg[:, chs["local"]:chs["local"] + nl] - feats["local"][0].permute(2, 3, 0, 1)).abs().max()),
                "local_proxy": float((fg[:, chs["local_proxy"]:chs["local_proxy"] + nl] - feats["local_proxy"][0].permute(2, 3, 0, 1)).abs().max()),
            }
            pg = fg[:, 0].reshape(O, -1)[:, ::DENSE_SUBSAMPLE].argmin(0)
            pc = feats["dense_sub"].argmin(1)
            iou_sum, iou_n = sharding.mask_iou_sums(pg, pc, O)
            parity = dict(max_abs_feature_diff=max(diffs.values()), per_branch=diffs, surrogate_mask_mean_iou=iou_sum / iou_n)

        value = metrics["frames"] / elapsed_max
        line = {
            "metric": "frames/sec, AOC-Net matching + calibration hot path (480p, 3 objects)",
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg.name}: {cfg.h}x{cfg.w} stride-4 maps{' (480p)' if (cfg.h, cfg.w) == (121, 213) else ''}, O={O} ({O - 1} objects + background), K={cfg.k} proxies, "
                                   f"C={C}, {cfg.frames}-frame clips, MEM_EVERY={mc.MEM_EVERY} (R=1..{1 + (cfg.frames - 2) // mc.MEM_EVERY}), "
                                   "20 Lloyd iterations, local windows [2..12]",
                       "sequences_per_gpu": n_streams, "frames_per_step": n_streams, "sharding": "sequences over ranks, no data-path collective",
                       "intra_frame_overlap": ("none" if args.no_overlap else "k-means chain on a side HIP stream" +
                                               ("" if args.no_pipeline else ", enqueued as soon as the pool it depends on is final "
                                                "(right after the previous frame's memory update)")),
                       "cu_reserve": (f"main streams masked off {args.cu_reserve} of 256 CUs (hipExtStreamCreateWithCUMask), left to the side-stream "
                                      "k-means chains" if args.cu_reserve > 0 else "none"),
                       "proxy_mode": ("NON-PARITY: adaptive proxies reused until the pool changes (one k-means per MEM_EVERY frames)"
                                      if args.reuse_proxies else "reference: the pool is re-clustered for every frame with that frame's initial rows"),
                       "dense_precision": ("fp16-split products (hi*hi + hi*lo + lo*hi), fp32 accumulate: fp32-equivalent; exact-fp32 take-over "
                                           "on overflow / soft labels" if args.dense == "split" else "exact fp32 MFMA")},
            "host_enqueue_ms_per_step": round(host_enqueue_s / args.steps * 1e3, 3),
            "exact_fp32_dense_run": exact,
            "roofline": roofline, "roofline_correlation_kernel": corr_roof, "roofline_kmeans_chain": km_roof, "cpu_baseline": cpu, "parity": parity, "kernels": kernels,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()                        # the other ranks wait for rank 0's report before tearing down
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
