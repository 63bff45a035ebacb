#!/usr/bin/env python3
"""Benchmark of the AOC-Net matching + calibration hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

(`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches itself under torch.distributed.run.)

A "step" is one frame of every in-flight synthetic video sequence (--streams per rank, default 2) through the whole hot path: label
prep, scipy-exact k-means proxies, proxy / dense / local matching, fg->bg, the proto-mask tensor, then the ten IA gates and four
conditioning blocks at the decoder's activation shapes.  Workload = BASELINE.json configs[1] (cfg2): 480p -> 121x213 stride-4 maps, 3
objects + background, K = 16 proxies, 60-frame clips whose reference pool holds one more frame every MEM_EVERY = 5 frames (R = 1..12),
as in the reference's eval loop (eval_manager_mm.py:309-361).  Inputs (feature maps, label maps, k-means initial rows, decoder
activations) are synthetic, seeded and resident in HBM before the timed region.

The frames of a clip are visited in GROUPS of MEM_EVERY consecutive frames (one pool state each) and the groups in a fixed interleaved
order, so that ANY number of steps sees small and large pools in proportion (the line prints the histogram of R actually timed);
inside a group everything happens as in the sequential loop: the first frame's k-means chain can only start once the pool is final,
the chains of the others are enqueued ahead.

Prints ONE compact JSON line on rank 0 as the LAST line of stdout (contract in the task statement; compact_line(): held under 4 KB by a CPU test):
whole-job frames/s, roofline objects measured with HIP events inside the run, and a CPU baseline (the oracle timed on the host cores, rank 0, N = 1
only).  Everything else -- notes, per-op tables, sweeps, the other configs' records -- goes to --details-file (default gpurun_out/bench_details.json),
whose path the line carries.  Optional legs (--extras: cfg3, cfg4, closed-loop, backbone, corr-sweep) are not started once --budget-s seconds of wall
time have passed; the default run takes about a minute.
"""
import argparse
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # main + dense + k-means stream per sequence: more than the default 4 hardware queues
import ctypes
import json
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import aoc_amd  # noqa: E402
from aoc_amd import eval_runner, hotpath, ops, sharding  # noqa: E402
from aoc_amd import synthetic as syn  # noqa: E402

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_F16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_f16, dense
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense

CONFIG_LEVELS = {"cfg3": [8, 16, 32]}     # BASELINE.json configs[2]: multi-level proxies


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
    """HIP-event timing of every call of selected aoc_amd.ops functions on the stream they are launched on (torch's current stream).
    Recording is switched on for the timed region only; the explicit ordering of the sequences' dense kernels (--dense-order) is applied always (warm-up
    and timed region schedule the same way)."""

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
        # raw hipEvent_t from a pre-created pool, recorded on the raw current stream: a torch.cuda.Event costs ~10 us of host time per
        # record (object construction + current-stream lookup), i.e. ~2 ms per step of this bench's own instrumentation
        self.hip = self.kernel_probe.hip
        self.hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
        self._pool = []
        self.streams = {}
        self.want_segments = False
        self.probe_every = 1
        self.segments = {}                                   # --segments: per stream, the event tuples of its frames in order

    def segment_summary(self):
        """Mean milliseconds of the consecutive pieces of a frame ON ITS STREAM (developer output, --segments): what comes before the dense
        op, the dense op, ..., the wait for the frame's proxies + the correlation launch, the gates, and the idle time to the next frame."""
        names = ["pre_dense (pools, local prep, split rows)", "dense_op", "dense_end_to_local", "local_pair", "planes + wait for the proxies", "correlation",
                 "proto_finish", "gates", "to_next_frame_on_this_stream"]
        acc = [[] for _ in names]
        for frames in self.segments.values():
            for i, ev in enumerate(frames):
                pts = list(ev) + ([frames[i + 1][0]] if i + 1 < len(frames) else [])
                for j in range(len(pts) - 1):
                    v = ctypes.c_float()
                    if self.hip.hipEventElapsedTime(ctypes.byref(v), pts[j], pts[j + 1]) == 0:
                        acc[j].append(v.value)
        return {n: round(float(np.mean(a)), 4) for n, a in zip(names, acc) if a}

    def timeline(self):
        """[(op, stream, start_ms, end_ms)] of every timed call, relative to the first recorded event (developer output: --dump-timeline)."""
        allrec = [(n, st, a, b) for n in self.names for (a, b), st in zip(self.records[n], self.streams.get(n, []))]
        if not allrec:
            return []
        base = allrec[0][2]
        out = []
        for n, st, a, b in allrec:
            va, vb = ctypes.c_float(), ctypes.c_float()
            if self.hip.hipEventElapsedTime(ctypes.byref(va), base, a) == 0 and self.hip.hipEventElapsedTime(ctypes.byref(vb), base, b) == 0:
                out.append((n, st, va.value, vb.value))
        return sorted(out, key=lambda r: r[2])

    def _event(self):
        if not self._pool:
            for _ in range(16384):
                e = ctypes.c_void_p()
                assert self.hip.hipEventCreate(ctypes.byref(e)) == 0
                self._pool.append(e)
        return self._pool.pop()

    def _record(self):
        e = self._event()
        assert self.hip.hipEventRecord(e, ops._stream()) == 0
        return e

    def install(self, meta_fns):
        self._pool.append(self._event())                     # fills the pool (setup)
        for n in self.names:
            fn = getattr(ops, n)
            self._orig[n] = fn

            def wrapped(*a, _fn=fn, _n=n, **k):
                dense = _n in self.probed
                if dense and self.dense_done is not None and self.serialize_dense:
                    # only one dense kernel fits per CU, so the dense kernels of the sequences run back to back anyway; making that
                    # order explicit keeps the queueing of one behind the other out of the timed interval
                    assert self.hip.hipStreamWaitEvent(ops._stream(), self.dense_done, 0) == 0
                if not self.enabled:
                    out = _fn(*a, **k)
                    if dense:
                        self.dense_done = self._record()
                    return out
                e0 = self._record()
                if dense:
                    self.kernel_probe.arm()
                out = _fn(*a, **k)
                e1 = self._record()
                if dense:
                    self.dense_done = e1
                self.records[_n].append((e0, e1))
                self.streams.setdefault(_n, []).append(ops._stream().value)
                self.meta[_n].append(meta_fns[_n](*a, **k) if _n in meta_fns else None)
                return out

            setattr(ops, n, wrapped)

    def gate_probes(self, gates, acts):
        plan = gates.plan(0, 0)
        return [[self._event(), self._event(), self._event() if name.startswith("CLB") else None, self._event() if name.startswith("CLB") else None]
                for (name, *_r) in plan]

    def gates_done(self, probes, gates, acts):
        st = ops._stream().value
        for (name, *_r), p, x in zip(gates.plan(0, 0), probes, acts):
            self.records["film_scale"].append((p[0], p[1]))
            self.streams.setdefault("film_scale", []).append(st)
            self.meta["film_scale"].append(dict(flops=float(x.numel()), bytes=2.0 * x.numel() * 4))
            if p[2] is not None:
                self.records["cond_gate_pool"].append((p[2], p[3]))
                self.streams.setdefault("cond_gate_pool", []).append(st)
                self.meta["cond_gate_pool"].append(dict(flops=2.0 * x.numel(), bytes=1.0 * x.numel() * 4))

    def frame_probes(self):
        """Six raw events for aoc_frame_enqueue's probe slots (dense op, correlation launch, local-matching launch) + the dense kernel probe."""
        self.kernel_probe.arm()
        return [self._event() for _ in range(6)]

    def frame_done(self, probes, n, m, n_proxy):
        st = ops._stream().value
        for name, (a, b), args in (("dense_match_min_split", probes[0:2], ("dense", m, n)), ("proxy_corr_min_records", probes[2:4], ("corr", m, n_proxy)),
                                   ("local_window_match", probes[4:6], None)):
            self.records[name].append((a, b))
            self.streams.setdefault(name, []).append(st)
            self.meta[name].append(self.frame_meta(*args) if args else None)

    def summary(self):
        out = {}
        for n in self.names:
            ms = []
            for a, b in self.records[n]:
                v = ctypes.c_float()
                if self.hip.hipEventElapsedTime(ctypes.byref(v), a, b) == 0:
                    ms.append(v.value)
            if ms:
                out[n] = dict(calls=len(ms), total_ms=float(np.sum(ms)), avg_ms=float(np.mean(ms)), meta=self.meta[n])
        return out


def group_order(n_groups):
    """Visiting order of the pool-state groups: smallest, largest, second smallest, second largest, ...: every window of an even
    number of groups has the clip's mean pool size, so any run length samples R in proportion."""
    order, lo, hi = [], 0, n_groups - 1
    while lo <= hi:
        order.append(lo)
        if hi != lo:
            order.append(hi)
        lo += 1
        hi -= 1
    return order


class ClipWorkload:
    """One synthetic sequence of a config, resident on the GPU.  The pool of frame t is emb[0::MEM_EVERY][:R(t)] with the clip's own
    (ground-truth) label maps, R(t) = 1 + (t - 1) // MEM_EVERY -- exactly what the sequential loop with the reference's memory policy
    builds -- so every pool state exists from the start and the frames can be visited group by group in any order."""

    def __init__(self, cfg, seed, device, mc, overlap=True, phase=0, start=0):
        self.cfg, self.mc, self.dev = cfg, mc, device
        # the k-means branch (a long chain of small launches) runs on a high-priority side stream,
        # concurrently with the MFMA-bound dense matching on the main stream
        self.side = torch.cuda.Stream(device=device, priority=-1) if overlap else None
        self.sides = [self.side]                           # bench --chain-streams: the chains of a group's batches alternate between these
        self.next_side = 0
        clip = syn.make_clip(cfg, seed)
        O = cfg.n_obj
        self.emb = torch.from_numpy(clip["emb"]).to(device)                                   # [T,h,w,C]
        self.lab_ids = clip["lab"]
        self.lab = torch.from_numpy(np.stack([syn.one_hot(l, O) for l in clip["lab"]])).to(device)   # [T,h,w,O]
        self.T = cfg.frames
        self.bias = torch.zeros(O, device=device)
        me = mc.MEM_EVERY
        self.pool_emb = self.emb[0:self.T - 1:me].contiguous()                               # pool frames 0, 5, 10, ...
        self.pool_lab = self.lab[0:self.T - 1:me].contiguous()
        self.rmax = self.pool_emb.shape[0]
        levels = mc.cluster_levels
        kmax = max(levels)
        # per-frame k-means initial rows (inputs): permutation(n_i)[:K_i] with the sticky K rule, per level
        self.init_rows = {}
        for t in range(1, self.T):
            R = self.R_of(t)
            counts = [int(sum((self.lab_ids[i * me] == o).sum() for i in range(R))) for o in range(O)]
            init = np.zeros((len(levels) * O, kmax), np.int32)
            host = []
            for li, k in enumerate(levels):
                rows = syn.kmeans_init_rows(seed * 100003 + t * 7 + li, counts, k)
                host.append(rows)
                for o, r in enumerate(rows):
                    if r is not None:
                        init[li * O + o, :len(r)] = r
            self.init_rows[t] = (torch.from_numpy(init).to(device), host)
        # visiting order: groups of MEM_EVERY frames that share a pool, groups interleaved; `phase` rotates it per sequence
        groups = [list(range(g * me + 1, min(g * me + me, self.T - 1) + 1)) for g in range((self.T - 2) // me + 1)]
        order = group_order(len(groups))
        if phase % 2:                                      # odd sequences walk (largest, smallest, ...): two sequences together balance any window
            order = [g for i in range(0, len(order), 2) for g in reversed(order[i:i + 2])]
        rot = (phase // 2) * 2 % len(order)
        order = order[rot:] + order[:rot]
        self.order = [t for g in order for t in groups[g]]
        # `start`: position of the (cyclic) walk at which this sequence begins.  Sequences that start inside a group cross their group
        # boundaries -- where a frame has to wait for a k-means chain that could not be enqueued ahead -- at different steps
        self.start = start % len(self.order)
        self.group_first = {groups[g][0] for g in range(len(groups))}
        self.group_of = {t: g for g in range(len(groups)) for t in groups[g]}
        self.groups = groups
        self.dense_state = {"capacity_frames": self.rmax}  # fp16 split records of the pool (one conversion per appended frame)
        self.ahead = {}                                    # frame -> adaptive proxies already enqueued on a side stream
        self.pool_event = None                             # recorded when the pool of the current group is final
        self.chains = mc.MEM_EVERY                         # k-means chains enqueued ahead (bench --chains)
        self.chain_lead = 1                                # batches enqueued ahead of the one in use (bench --chain-lead)
        self.chain_plan = None                             # batch sizes of a group's chains, e.g. [1, 2, 2] (bench --chain-plan)
        self.dense_stream = None                           # CU-masked stream for the dense kernel alone
        self.reuse_proxies = False                         # non-parity mode: one k-means per pool state
        self.bank = None                                   # non-parity mode: hotpath.IncrementalProxyBank (one clustering per pool FRAME)
        self.cached_ahead = None
        self.batch_gates = False                           # the gates of a frame as ONE C call (aoc_gates_enqueue)
        self.runner = None                                 # hotpath.FrameRunner: one C call per frame (aoc_frame_enqueue); None = the Python orchestrator
        self.timer = None
        self.pool_gen = 0
        self.r_hist = {}
        self.count_r = False
        self.reset()

    def R_of(self, t):
        return 1 + (t - 1) // self.mc.MEM_EVERY

    def reset(self):
        self.pos = self.start
        if self.runner is not None:
            self.runner.reset()
        self.dense_state["frames"] = 0
        self.dense_state.pop("ref_pool", None)
        self.cached_ahead = None
        self.ahead.clear()
        self._enter_frame()

    @property
    def t(self):
        return self.order[self.pos]

    @property
    def R(self):
        return self.R_of(self.t)

    def enable_incremental(self):
        """NON-PARITY mode: every pool frame is clustered once, on its own rows; frames match against the union of the
        per-frame code books.  All pool frames are clustered here (setup); the walk re-clusters a group's newest pool frame when it
        enters the group, which is what the sequential loop pays per pool update."""
        self.bank = hotpath.IncrementalProxyBank(self.mc, self.cfg.n_obj, self.cfg.c, self.rmax, self.dev)
        for r in range(self.rmax):
            self.bank.append(self.pool_emb[r], self.pool_lab[r], self._bank_init(r), self.side)

    def _bank_init(self, r):
        O, K = self.cfg.n_obj, self.mc.cluster_levels[0]
        counts = [int((self.lab_ids[r * self.mc.MEM_EVERY] == o).sum()) for o in range(O)]
        rows = syn.kmeans_init_rows(777 + r, counts, K)
        init = np.zeros((O, K), np.int32)
        for o, rr in enumerate(rows):
            if rr is not None:
                init[o, :len(rr)] = rr
        return torch.from_numpy(init).to(self.dev)

    def _enter_frame(self):
        """Book-keeping when the walk reaches a frame: at the first frame of a group the pool has just received its newest frame
        (eval_manager_mm.py:309-312): that frame's split records are converted again and nothing that depends on the pool may have
        been enqueued earlier."""
        if self.t in self.group_first:
            self.pool_gen += 1
            if self.runner is not None:
                self.runner.pool_changed(self.R - 1)
            self.dense_state["frames"] = min(self.dense_state.get("frames", 0), self.R - 1)
            self.dense_state.pop("ref_pool", None)
            self.cached_ahead = None
            self.ahead.clear()
            self.pool_event = torch.cuda.Event()
            self.pool_event.record()
            if self.bank is not None:                      # the newest pool frame joins the bank: ONE single-frame clustering per pool update
                r = self.R - 1
                if not hasattr(self, "_bank_inits"):
                    self._bank_inits = [self._bank_init(i) for i in range(self.rmax)]
                self.bank.append(self.pool_emb[r], self.pool_lab[r], self._bank_inits[r], self.side, wait_event=self.pool_event, slot=r)

    def refs(self):
        return self.pool_emb[:self.R], self.pool_lab[:self.R]

    def next_in_group(self):
        """Frames after the current one that see the same pool (they follow it directly in the walk)."""
        g = self.groups[self.group_of[self.t]]
        return [u for u in g if u > self.t]

    def pretouch(self, gates, acts, dense_precision, pipeline):
        """Setup, not a benchmark step: one frame at EVERY pool size the clip will reach, largest first, so that torch's
        caching allocator already owns a block for every request of the timed region (otherwise each pool growth calls
        hipMalloc, which synchronises the device)."""
        t_of = {self.R_of(t): t for t in range(1, self.T)}
        for R in range(self.rmax, 0, -1):
            t = t_of[R]
            self.dense_state["frames"] = 0
            self.dense_state.pop("ref_pool", None)
            self.pool_gen += 1
            if self.runner is not None:
                self.runner.reset()
            ref_emb, ref_lab = self.pool_emb[:R], self.pool_lab[:R]
            init = self.init_rows[t][0]
            ev = torch.cuda.Event()
            ev.record()
            if self.side is not None and pipeline:
                ahead = hotpath.launch_cluster_proxies(self.mc, ref_emb, ref_lab, init, self.side, wait_event=ev)
                for nb in (sorted({b for b in self.chain_plan if b > 1}) if self.chain_plan else range(2, self.chains)):   # batched chains of every size the run can ask for
                    hotpath.launch_cluster_proxies_batch(self.mc, ref_emb, ref_lab, [init] * nb, self.side, wait_event=ev)
                if self.runner is not None and dense_precision == "split" and self.dense_stream is None:
                    feat, head = self.runner(ref_emb, ref_lab, self.emb[t - 1], self.lab[t - 1], self.emb[t], self.bias, ahead, pool_key=self.pool_gen)
                feat, head, _ = hotpath.proto_mask_features(self.mc, ref_emb, ref_lab, self.emb[t - 1], self.lab[t - 1], self.emb[t], self.bias,
                                                            cluster_ahead=ahead, dense_state=self.dense_state, dense_precision=dense_precision,
                                                            dense_stream=self.dense_stream)
            else:
                feat, head, _ = hotpath.proto_mask_features(self.mc, ref_emb, ref_lab, self.emb[t - 1], self.lab[t - 1], self.emb[t], self.bias,
                                                            cluster_state=dict(init_rows=init), side_stream=self.side,
                                                            dense_state=self.dense_state, dense_precision=dense_precision)
            gates(acts, head)
            torch.cuda.synchronize()
        self.reset()

    def advance(self):
        if self.count_r:
            self.r_hist[self.R] = self.r_hist.get(self.R, 0) + 1
        self.pos = (self.pos + 1) % len(self.order)
        self._enter_frame()


def make_activations(gates, O, h, w, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return [torch.randn(O, c, hh, ww, generator=g).to(device) for (_, c, hh, ww, _) in gates.plan(h, w)]


def _launch_batch(wl, frames):
    ref_emb, ref_lab = wl.refs()
    side = wl.sides[wl.next_side % len(wl.sides)]          # independent chains (same pool, different initial rows) may run side by side
    wl.next_side += 1
    if len(frames) == 1:
        wl.ahead[frames[0]] = hotpath.launch_cluster_proxies(wl.mc, ref_emb, ref_lab, wl.init_rows[frames[0]][0], side, wait_event=wl.pool_event)
    else:
        outs = hotpath.launch_cluster_proxies_batch(wl.mc, ref_emb, ref_lab, [wl.init_rows[f][0] for f in frames], side, wait_event=wl.pool_event)
        for f, a in zip(frames, outs):
            wl.ahead[f] = a


HOST_S = dict(chain_launch=0.0, frame_call=0.0, gates_call=0.0, frames=0)      # --segments: host seconds spent inside the three enqueue calls


def _launch_batch_timed(wl, todo):
    t0 = time.perf_counter()
    _launch_batch(wl, todo)
    HOST_S["chain_launch"] += time.perf_counter() - t0


def launch_chains(wl, done=None):
    """Enqueue the k-means chain of the current frame and of the following frames of its group (they all see the pool as it is now)
    on the side stream.  With a plan (--chain-plan, e.g. 1,2,2): the group's frames are cut into batches of those sizes, one chain per
    batch; the batch of the current frame and the one after it are enqueued (one batch of lead).  Without: the current frame alone
    (it is needed first), the next --chains - 1 frames batched into one chain."""
    t = wl.t
    if wl.chain_plan:
        g = wl.groups[wl.group_of[t]]
        cuts, i = [], 0
        for b in wl.chain_plan:
            if i < len(g):
                cuts.append(g[i:i + b])
                i += b
        if i < len(g):
            cuts.append(g[i:])
        k = next(j for j, c in enumerate(cuts) if t in c)
        for c in cuts[k:k + 1 + max(wl.chain_lead, len(wl.sides))]:
            todo = [f for f in c if f >= t and f != done and f not in wl.ahead]
            if todo:
                _launch_batch_timed(wl, todo)
        return
    rest = wl.next_in_group()[:max(0, wl.chains - 1)]
    _launch_batch_timed(wl, [t])
    if rest:
        _launch_batch_timed(wl, rest)


def frame_step(wl, gates, acts, dense_precision="split", pipeline=True, defer_corr=False):
    """One frame of one sequence; returns (feat, gate outputs, pending correlation or None)."""
    ref_emb, ref_lab = wl.refs()
    t = wl.t
    seg = probes = None
    if wl.bank is not None:
        feat, head, aux = hotpath.proto_mask_features(wl.mc, ref_emb, ref_lab, wl.emb[t - 1], wl.lab[t - 1], wl.emb[t], wl.bias,
                                                      cluster_ahead=wl.bank.handle(ref_lab), dense_state=wl.dense_state, dense_precision=dense_precision,
                                                      dense_stream=wl.dense_stream, defer_correlation=defer_corr)
    elif wl.side is not None and pipeline:
        # the k-means chain of a frame only depends on the pool: the chains of all frames of a group are enqueued on the side stream as
        # soon as the group's pool is final and run under the other work of the frames before them
        if wl.reuse_proxies and wl.cached_ahead is not None and wl.cached_ahead.R == wl.R:
            # NON-PARITY mode: the adaptive proxies computed for this pool are reused until the pool changes
            ahead = wl.cached_ahead
        else:
            if t not in wl.ahead:
                launch_chains(wl)
            ahead = wl.ahead.pop(t)
            wl.cached_ahead = ahead if wl.reuse_proxies else None
        nxt = wl.next_in_group()
        if wl.chain_plan:
            if not wl.reuse_proxies and nxt:
                launch_chains(wl, done=t)                   # keeps one batch of lead (no-op when it is already enqueued)
        elif not wl.reuse_proxies and nxt and nxt[0] not in wl.ahead:
            wl.ahead[nxt[0]] = hotpath.launch_cluster_proxies(wl.mc, ref_emb, ref_lab, wl.init_rows[nxt[0]][0], wl.side, wait_event=wl.pool_event)
        if wl.runner is not None and dense_precision == "split" and wl.dense_stream is None and not defer_corr:
            # ONE C call for the frame (aoc_frame_enqueue); the op timings the line reports come from events the call records itself
            tm = wl.timer
            # the event brackets are placed on every --probe-every-th frame of a sequence (on every frame they cost 2.6 % of the throughput:
            # ~40 event records per frame on the stream); the two sequences are probed on different steps
            wl.n_probe_tick = getattr(wl, "n_probe_tick", getattr(wl, "probe_phase", 0)) + 1
            sample = tm is not None and tm.enabled and wl.n_probe_tick % max(tm.probe_every, 1) == 0
            probes = tm.frame_probes() if sample else None
            seg_begin = tm._record() if (probes is not None and tm.want_segments) else None
            h0 = time.perf_counter()
            feat, head = wl.runner(ref_emb, ref_lab, wl.emb[t - 1], wl.lab[t - 1], wl.emb[t], wl.bias, ahead, pool_key=wl.pool_gen, probes=probes)
            HOST_S["frame_call"] += time.perf_counter() - h0
            HOST_S["frames"] += 1
            if probes is not None:
                tm.frame_done(probes, ref_emb.shape[0] * ref_emb.shape[1] * ref_emb.shape[2], wl.emb[t].shape[0] * wl.emb[t].shape[1], ahead.table.shape[0])
                if seg_begin is not None:
                    seg = [seg_begin, probes[0], probes[1], probes[4], probes[5], probes[2], probes[3], tm._record()]
            aux = dict(pending_correlation=None)
        else:
            feat, head, aux = hotpath.proto_mask_features(wl.mc, ref_emb, ref_lab, wl.emb[t - 1], wl.lab[t - 1], wl.emb[t], wl.bias,
                                                          cluster_ahead=ahead, dense_state=wl.dense_state, dense_precision=dense_precision,
                                                          dense_stream=wl.dense_stream, defer_correlation=defer_corr)
    else:
        feat, head, aux = hotpath.proto_mask_features(wl.mc, ref_emb, ref_lab, wl.emb[t - 1], wl.lab[t - 1], wl.emb[t], wl.bias,
                                                      cluster_state=dict(init_rows=wl.init_rows[t][0]), side_stream=wl.side,
                                                      dense_state=wl.dense_state, dense_precision=dense_precision, defer_correlation=defer_corr)
    if wl.batch_gates:
        tm = wl.timer
        gp = tm.gate_probes(gates, acts) if (tm is not None and tm.enabled and (probes is not None or wl.runner is None)) else None
        h0 = time.perf_counter()
        outs = gates.forward_batched(acts, head, slot=id(wl), probes=gp)
        HOST_S["gates_call"] += time.perf_counter() - h0
        if gp is not None:
            tm.gates_done(gp, gates, acts)
            if seg is not None:
                tm.segments.setdefault(ops._stream().value, []).append(seg + [tm._record()])
    else:
        outs = gates(acts, head)
    wl.advance()                                           # the walk moves on (a new group = a new pool state starts here)
    if wl.bank is None and wl.side is not None and pipeline and wl.t not in wl.ahead and not (wl.reuse_proxies and wl.cached_ahead is not None and wl.cached_ahead.R == wl.R):
        launch_chains(wl)                                  # the next frame's pool is final now: its chain starts under the other sequences' work
    return feat, outs, aux["pending_correlation"]


def _block_weights(mod):
    sd = {k: v.detach().cpu() for k, v in mod.state_dict().items()}
    return {"CL_1.phi_w": sd["CL_1.phi_layer.weight"].reshape(-1), "CL_1.phi_b": sd["CL_1.phi_layer.bias"],
            "CL_1.mlp_w": sd["CL_1.mlp_layer.weight"], "CL_1.mlp_b": sd["CL_1.mlp_layer.bias"],
            "CL_2.mlp_w": sd["CL_2.mlp_layer.weight"], "CL_2.mlp_b": sd["CL_2.mlp_layer.bias"],
            "CL_3.mlp_w": sd["CL_3.mlp_layer.weight"], "CL_3.mlp_b": sd["CL_3.mlp_layer.bias"],
            "mlp_w": sd["mlp_layer.weight"], "mlp_b": sd["mlp_layer.bias"]}


DENSE_SUBSAMPLE = 16             # --cpu-dense-subsample
CPU_BASELINE_R = 6             # pool frames of the CPU sample: the clip's average is 6.4


def cpu_baseline(cfg, mc, seed, gates, acts_cpu, runs=5):
    """The oracle (a port of the reference path, proved equal to it on the golden vectors) timed on the host cores for ONE frame of
    the same workload at R = 6 pool frames (the clip's average is 6.4), bounded to a few tens of seconds: the dense branch (linear in
    query pixels) runs on every DENSE_SUBSAMPLE-th query pixel and is scaled by that factor; each distinct calibration gate shape is
    timed once per run and multiplied by its count.  Protocol of BASELINE.md section 3: one warm-up run, then `runs` timed runs of the
    whole frame; the reported rate is the MEDIAN frame time (the minimum is kept in the details).  A second, cheap frame at R = 1
    provides the per-branch features for the parity spot check."""
    from oracle import calibration as ocal
    from oracle import matching as om
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    me = mc.MEM_EVERY
    clip = syn.make_clip(cfg, seed, frames=(CPU_BASELINE_R - 1) * me + 3)
    O = cfg.n_obj
    emb = torch.from_numpy(clip["emb"])
    lab = torch.from_numpy(np.stack([syn.one_hot(l, O) for l in clip["lab"]]))
    levels = mc.cluster_levels
    cn = levels if mc.CLUSTER_LEVELS else levels[0]
    bias = torch.zeros(O)
    mld = list(mc.MODEL_MULTI_LOCAL_DISTANCE)

    def rows_for(ref_ids, s):
        counts = [int(sum((clip["lab"][i] == o).sum() for i in ref_ids)) for o in range(O)]
        rows = [syn.kmeans_init_rows(seed * 100003 + s + li, counts, k) for li, k in enumerate(levels)]
        return rows if mc.CLUSTER_LEVELS else rows[0]

    # ---- the timed frame: pool = frames 0, 5, .., 25, previous frame 26, query 27
    ref_ids = [i * me for i in range(CPU_BASELINE_R)]
    tq = ref_ids[-1] + 2
    refs, labs = [emb[i] for i in ref_ids], [lab[i] for i in ref_ids]
    ref_flat = torch.cat([r.reshape(-1, cfg.c) for r in refs])
    lab_flat = torch.cat([l.reshape(-1, O) for l in labs])
    q_sub = emb[tq].reshape(-1, cfg.c)[::DENSE_SUBSAMPLE]
    ref_e = [e.permute(2, 0, 1).unsqueeze(0) for e in refs]
    ref_l = [l.permute(2, 0, 1).unsqueeze(1) for l in labs]
    init = rows_for(ref_ids, 1)
    weights = {}
    for (name, c, hh, ww, extra) in gates.plan(cfg.h, cfg.w):
        mod = getattr(gates, name)
        weights[name] = _block_weights(mod) if name.startswith("CLB") else {k: v.detach().cpu() for k, v in mod.state_dict().items()}

    def one_frame():
        tm = {}

        def timed(key, fn):
            t0 = time.perf_counter()
            out = fn()
            tm[key] = tm.get(key, 0.0) + time.perf_counter() - t0
            return out

        timed("dense", lambda: om.proto_transform(om.nearest_neighbor_features_per_object(ref_flat, q_sub, lab_flat).squeeze(-1), bias.view(1, -1)))
        tm["dense"] *= DENSE_SUBSAMPLE
        timed("cluster", lambda: om.global_matching_for_eval_cluster(refs, emb[tq], labs, 4, bias, init_rows=init, cluster_num=cn))
        timed("local", lambda: om.local_matching(emb[tq - 1], emb[tq], lab[tq - 1], bias, mld))
        head, ref_pos, _, prev_pos, _ = timed("pool", lambda: ocal.attention_head_for_eval_p_m(
            ref_e, ref_l, emb[tq - 1].permute(2, 0, 1).unsqueeze(0).expand(O, -1, -1, -1), lab[tq - 1].permute(2, 0, 1).unsqueeze(1), mc.MODEL_EPSILON))
        timed("proxy", lambda: om.global_matching_for_eval_proxy(ref_pos, emb[tq], labs, 4, bias))
        timed("local_proxy", lambda: om.local_matching(torch.matmul(lab[tq - 1], prev_pos), emb[tq], lab[tq - 1], bias, mld))
        t_match = sum(tm.values())
        t_cal, seen = 0.0, {}
        for (name, c, hh, ww, extra), x in zip(gates.plan(cfg.h, cfg.w), acts_cpu):
            key = (name.startswith("CLB"), c, hh, ww, extra)
            if key not in seen:
                t0 = time.perf_counter()
                if name.startswith("CLB"):
                    ocal.conditioning_block(x, head, weights[name], mc.BETA_PERCENTAGE)
                else:
                    hd = head
                    if extra:
                        px = x.mean(dim=(2, 3))
                        hd = torch.cat([head, px.sum(0, keepdim=True) - px], 1)
                    ocal.ia_gate(x, hd, weights[name]["IA.weight"], weights[name]["IA.bias"])
                seen[key] = time.perf_counter() - t0
            t_cal += seen[key]
        return tm, t_match, t_cal

    one_frame()                                            # warm-up (allocator, thread pool, page faults)
    samples = [one_frame() for _ in range(max(1, runs))]
    order = sorted(range(len(samples)), key=lambda i: samples[i][1] + samples[i][2])
    tm, t_match, t_cal = samples[order[len(order) // 2]]
    frame_s = [round(s[1] + s[2], 4) for s in samples]

    # ---- parity frame (R = 1): per-branch features of the oracle, untimed
    feats = {}
    rows1 = rows_for([0], 2)
    q1_sub = emb[1].reshape(-1, cfg.c)[::DENSE_SUBSAMPLE]
    feats["dense_sub"] = om.proto_transform(om.nearest_neighbor_features_per_object(emb[0].reshape(-1, cfg.c), q1_sub, lab[0].reshape(-1, O)).squeeze(-1),
                                            bias.view(1, -1))
    feats["cluster"] = om.global_matching_for_eval_cluster([emb[0]], emb[1], [lab[0]], 4, bias, init_rows=rows1, cluster_num=cn)
    feats["local"] = om.local_matching(emb[0], emb[1], lab[0], bias, mld)
    e0 = emb[0].permute(2, 0, 1).unsqueeze(0)
    l0 = lab[0].permute(2, 0, 1).unsqueeze(1)
    _, rp, _, pp, _ = ocal.attention_head_for_eval_p_m([e0], [l0], e0.expand(O, -1, -1, -1), l0, mc.MODEL_EPSILON)
    feats["proxy"] = om.global_matching_for_eval_proxy(rp, emb[1], [lab[0]], 4, bias)
    feats["local_proxy"] = om.local_matching(torch.matmul(lab[0], pp), emb[1], lab[0], bias, mld)
    return feats, rows1, (emb[:2], lab[:2]), tm, t_match, t_cal, threads, frame_s


def pmc_traffic_bytes(relpath):
    """HBM-side bytes per launch from a committed rocprofv3 --pmc summary under profiles/ (separate FETCH_SIZE / WRITE_SIZE passes, per-dispatch
    averages in KB; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes, MI355X_MICROARCH.md): FETCH x 2 + WRITE.  The file is read at
    bench time (counters cannot be collected inside the timed run); None when it is missing or has no such lines."""
    import re
    try:
        text = open(os.path.join(ROOT, relpath)).read()
    except OSError:
        return None
    f = re.search(r"^FETCH_SIZE\s+n=\s*\d+\s+avg=\s*([0-9.e+]+)", text, re.M)
    w = re.search(r"^WRITE_SIZE\s+n=\s*\d+\s+avg=\s*([0-9.e+]+)", text, re.M)
    if not f or not w:
        return None
    return (2.0 * float(f.group(1)) + float(w.group(1))) * 1024.0


COMPACT_LIMIT = 4096           # bytes: the driver keeps an 8 KB tail of stdout; round 5's 23 KB line did not parse
SCHEMA = 6                     # round 6: compact line + details file; host_enqueue_ms_per_step has its rounds-1-4 meaning again


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None} if isinstance(d, dict) else None


def compact_line(full, details_file):
    """The ONE line bench.py prints: the contract's keys, the roofline objects reduced to numbers, and a pointer to the details file that holds
    everything else (notes, isolated sweeps, per-op tables, other configs).  Pure function of the full record so that a CPU test can hold it to
    COMPACT_LIMIT bytes (tests/test_host_logic.py::test_bench_line_is_compact)."""
    roof_keys = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "in_run_frac", "pipe_frac", "frames_per_launch",
                 "best_frac", "best_frames_per_launch")
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                      "dtype", "data")}
    cfg = full.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "sequences_per_gpu", "R_mean", "proxy_mode"))
    for key in ("roofline", "roofline_correlation", "roofline_kmeans_chain", "roofline_film_scale"):
        r = full.get(key)
        if isinstance(r, dict):
            r = {k: (r[k] if k in r else None) for k in roof_keys if k in r or k == "traffic"}
            if isinstance(r.get("kernel"), str):
                r["kernel"] = r["kernel"].split(" (")[0][:48]
        line[key] = r
    cpu = full.get("cpu_baseline")
    line["cpu_baseline"] = None if cpu is None else dict(_pick(cpu, ("value", "unit", "cores", "kind", "runs")), sample=str(cpu.get("sample", ""))[:200])
    par = full.get("parity")
    line["parity"] = None if par is None else _pick(par, ("max_abs_feature_diff", "surrogate_mask_mean_iou"))
    for k in ("exact_fp32_value", "cfg3_value", "cfg4_value", "closed_loop_value", "timed_regions", "host_enqueue_ms_per_step",
              "host_enqueue_cpu_ms_per_step", "ranks_seen", "frames_per_rank", "imbalance", "wall_s", "skipped"):
        if full.get(k) is not None:
            line[k] = full[k]
    line["schema"] = SCHEMA
    line["details_file"] = details_file
    text = json.dumps(line)
    if len(text) >= COMPACT_LIMIT:                         # never print a line the driver cannot parse: drop the optional objects, longest first
        for k in ("roofline_film_scale", "roofline_kmeans_chain", "frames_per_rank", "skipped", "roofline_correlation"):
            line.pop(k, None)
            text = json.dumps(line)
            if len(text) < COMPACT_LIMIT:
                break
    assert len(text) < COMPACT_LIMIT and "\n" not in text, len(text)
    return line


def _drain_stdout():
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def finish(report, use_dist):
    """The JSON line has to be the LAST line of the job's stdout (that is what the driver parses), and RCCL announces itself there ("Librccl path : ...")
    through a buffered C stream that is otherwise flushed at exit, after the report.  So: every rank drains its C-level stdout, the group is torn down,
    the ranks drain again, and only then does rank 0 print (after a short pause with several ranks: their output reaches the launcher through pipes)."""
    _drain_stdout()
    if use_dist:
        torch.distributed.barrier()                        # the other ranks wait for rank 0 before tearing down
        torch.distributed.destroy_process_group()
        _drain_stdout()
    if report is not None:
        if use_dist and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            time.sleep(0.5)
        print(report, flush=True)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def correlation_roofline(cfg, mc, dev, batches=(1, 4, 16, 32), reps=20):
    """The correlation kernel by itself: B distinct frames per aoc_proxy_corr_min_records launch (the product path's entry
    point: the queries as the tile-major split records the dense kernel consumes), B in `batches`, each launch bracketed by its own pair of
    HIP events on an otherwise idle GPU (median of `reps`; the argument marshalling is done once, outside the bracket -- the bracket holds the
    C call, i.e. the tile-table packing, the kernel and the 4 us gated exact-fp32 take-over kernel)."""
    O, C, hw = cfg.n_obj, cfg.c, cfg.h * cfg.w
    levels = mc.cluster_levels
    L, kmax = len(levels), max(levels)
    n_ad = L * O * 2 * kmax
    ch = hotpath.channel_slices(mc)
    stride = mc.proto_channels * hw
    sb, ss, so = [], [], []
    for l, k in enumerate(levels):
        for o in range(O):
            for f in range(2):
                sb.append(((l * O + o) * 2 + f) * kmax)
                ss.append(k)
                so.append(o * stride + (ch["cluster"] + 2 * l + f) * hw)
    for o in range(O):
        sb.append(n_ad + o)
        ss.append(1)
        so.append(o * stride + ch["proxy"] * hw)
    n_set = len(sb)
    rng = np.random.RandomState(0)
    frames = []
    for _ in range(max(batches)):
        q = torch.from_numpy(syn.fresh_embedding(rng, cfg.h, cfg.w, C)).to(dev).reshape(hw, C)
        table = torch.from_numpy((np.maximum(rng.randn(n_ad + O, C), 0) * 0.1).astype(np.float32)).to(dev)
        frames.append((q, table, table.pow(2).sum(1), torch.zeros(n_set, device=dev), torch.empty(O, mc.proto_channels, cfg.h, cfg.w, device=dev)))
    algo = hw * C * 4 + (n_ad + O) * C * 4 + 4 * hw * n_set
    splits = [ops.split_rows(f[0], tiled=True) for f in frames]           # what the product path has made for the dense kernel anyway
    # the conversion of one frame's query into tile-major split records (aoc_split_rows_tiled): shared with the dense kernel in the product, i.e.
    # outside the correlation launch; timed here so that the line can also state the bracket that includes it
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    torch.cuda.synchronize()
    for i in range(reps):
        e0[i].record()
        ops.split_rows(frames[0][0], tiled=True)
        e1[i].record()
    torch.cuda.synchronize()
    split_ms = float(np.median([e0[i].elapsed_time(e1[i]) for i in range(reps)]))
    out = []
    for b in batches:
        launch = ops.proxy_corr_min_records([(f[0], sp, f[1], f[2], f[3], f[4]) for f, sp in zip(frames[:b], splits)], sb, ss, so, True,
                                            prepare_only=True)
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
        for i in range(reps):
            ev0[i].record()
            launch()
            ev1[i].record()
        torch.cuda.synchronize()
        ms = float(np.median([ev0[i].elapsed_time(ev1[i]) for i in range(reps)]))
        gbs = b * algo / (ms * 1e-3) / 1e9
        out.append(dict(frames_per_launch=b, avg_launch_ms=round(ms, 4), algorithmic_bytes_per_launch=b * algo, achieved=round(gbs, 1),
                        frac=round(gbs / PEAK_HBM_GBS, 4), query_split_ms_per_frame=round(split_ms, 4),
                        frac_including_query_split=round(b * algo / ((ms + b * split_ms) * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)))
    return out


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
    ap.add_argument("--cpu-dense-subsample", type=int, default=16,
                    help="cpu_baseline: time the dense branch of the oracle (linear in query pixels) on every N-th query pixel and scale by N; 1 = the whole "
                         "frame, ~21 s per run on 32 threads (rounds 4-5, one run); the default 16 bounds one run to ~4.5 s so that the protocol's "
                         "warm-up + --cpu-runs timed runs fit the default run")
    ap.add_argument("--cpu-runs", type=int, default=5, help="cpu_baseline: timed runs of the frame after one warm-up run; the MEDIAN is reported (BASELINE.md section 3)")
    ap.add_argument("--exact-steps", type=int, default=10,
                    help="steps of the informational second region with the exact-fp32 dense kernel (reported as exact_fp32_dense_run; 0 = skip)")
    ap.add_argument("--cu-reserve", type=int, default=32,
                    help="keep the main streams off this many CUs (HIP CU mask) so the side-stream k-means chain always finds free CUs")
    ap.add_argument("--chains", type=int, default=3,
                    help="frames whose k-means is enqueued right after a pool update (1 = only the next frame; the others are batched into one chain)")
    ap.add_argument("--chain-plan", default=None,
                    help="batch sizes of the k-means chains of a group of MEM_EVERY frames, e.g. 1,2,2 (overrides --chains).  Default: 2,3 for cfg2 "
                         "(340-342 frames/s against 332-334 with --chains 3, three runs each; cfg3 / cfg4 are 2 %% slower with it), none otherwise; "
                         "'none' = the --chains schedule")
    ap.add_argument("--probe-every", type=int, default=4,
                    help="HIP-event brackets (in-run op timings of the roofline objects) on every N-th frame of a sequence; 1 = every frame (costs 2.6 %% frames/s)")
    ap.add_argument("--segments", action="store_true", help="developer output: mean time of the consecutive pieces of a frame on its stream (key frame_segments_ms)")
    ap.add_argument("--host-profile", default="", help="developer output: cProfile of the drained-queue enqueue steps (host_enqueue_ms_per_step), top entries to this file")
    ap.add_argument("--dump-timeline", default="", help="developer output: write the timed ops' (name, stream, start, end) to this JSON file")
    ap.add_argument("--chain-streams", type=int, default=1, help="side streams per sequence for its k-means chains (with --chain-plan: the batches of a group run side by side)")
    ap.add_argument("--chain-lead", type=int, default=1, help="with --chain-plan: batches enqueued ahead of the one in use")
    ap.add_argument("--dense-stream", dest="mask_main", action="store_false",
                    help="put only the dense kernel under the CU mask (on its own stream) instead of the whole main stream")
    ap.add_argument("--reuse-proxies", action="store_true",
                    help="NON-PARITY mode (SURVEY 8f-3): cluster the pool once per pool update instead of once per frame; the JSON "
                         "line then says so in config.proxy_mode and is not comparable with the default")
    ap.add_argument("--incremental-proxies", action="store_true",
                    help="NON-PARITY mode (SURVEY 8f-3, hotpath.IncrementalProxyBank): cluster every reference frame once, when it joins the pool, and match "
                         "against the union of the per-frame code books; config.proxy_mode says so and the line is not comparable with the default")
    ap.add_argument("--batch-corr", action="store_true",
                    help="ONE batched correlation launch per step for the frames of all in-flight sequences instead of one launch per sequence and "
                         "frame (measured slower with 2 sequences in flight: the shared launch makes the streams wait for each other every step)")
    ap.add_argument("--dense-order", action="store_true",
                    help="order the in-flight sequences' dense kernels explicitly (one after the other: the kernel's live timing then excludes "
                         "the time it shares the chip with the other sequence's dense kernel -- 0.24 instead of 0.22 of the fp16 peak -- at 2 %% "
                         "fewer frames/s: 325 against 332)")
    ap.add_argument("--no-dense-order", action="store_true", help="(default since round 3; kept for old command lines)")
    ap.add_argument("--no-stagger", dest="stagger", action="store_false",
                    help="start every in-flight sequence at a group boundary (default: sequence s starts s * MEM_EVERY / streams frames into its first group, "
                         "so that the sequences need their un-prefetchable first-of-group k-means chains at different steps)")
    ap.add_argument("--python-frames", action="store_true",
                    help="drive every frame through hotpath.proto_mask_features (the individual C entry points, ~45 ctypes calls per frame) instead of ONE "
                         "aoc_frame_enqueue call per frame (default since round 4; bit-identical results)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="start a frame's k-means chain with the frame instead of as soon as its pool is final")
    ap.add_argument("--no-overlap", action="store_true", help="run the k-means branch on the main stream (no intra-frame stream overlap)")
    ap.add_argument("--eval-sharded", action="store_true",
                    help="BASELINE.json configs[4]: run the sequence-sharded evaluation (eval_runner) over the ranks instead of the cfg2 step loop; "
                         "a fixed sequence set (strong scaling), --eval-scale of the 30 + 507 sequences")
    ap.add_argument("--eval-scale", type=float, default=0.03)
    ap.add_argument("--min-region-s", type=float, default=1.0,
                    help="the timed region of --steps steps is repeated until this many seconds have been timed (at most 15 regions); the line reports the MEDIAN region")
    ap.add_argument("--extras", default="cfg3,cfg4",
                    help="comma list of the optional legs, each written to the details file: cfg3, cfg4 (short regions of the other single-GPU configs, child "
                         "processes, N = 1 only), closed-loop (the fixed-set sequence-sharded evaluation, any N), backbone (tools/backbone_e2e.py), "
                         "corr-sweep (the correlation kernel alone at 1 / 4 / 16 / 32 frames per launch); 'all' / 'none'.  A leg is skipped (and listed under "
                         "`skipped`) once --budget-s seconds of wall time have passed")
    ap.add_argument("--no-extras", action="store_true", help="same as --extras none")
    ap.add_argument("--budget-s", type=float, default=100.0, help="wall-time budget of the whole run in seconds: optional legs are not STARTED after it (0 = no limit)")
    ap.add_argument("--details-file", default=os.path.join("gpurun_out", "bench_details.json"),
                    help="where rank 0 writes the full record (notes, per-op tables, sweeps, other configs); the printed line carries the path")
    ap.add_argument("--strong-scale", type=float, default=0.12, help="share of the 537-sequence evaluation set used for the strong-scaling figure (64 sequences)")
    ap.add_argument("--eval-lanes", type=int, default=4,
                    help="sequences in flight per rank in the closed evaluation loop, each on its own HIP stream (one MI355X, 16-sequence set at the "
                         "end of round 3: 2 / 3 / 4 / 5 / 6 / 8 / 10 lanes = 226 / 241 / 245-251 / 242 / 239 / 234 / 225 frames/s; with the slower "
                         "kernels of the round's first half six lanes were best: 158 / 165 / 174 / 176 / 181 / 174 / 176)")
    args = ap.parse_args()
    t_start = time.perf_counter()
    phases = {}
    global DENSE_SUBSAMPLE
    DENSE_SUBSAMPLE = max(1, args.cpu_dense_subsample)
    extras = set() if (args.no_extras or args.extras == "none") else set(x.strip() for x in args.extras.split(",") if x.strip())
    if "all" in extras:
        extras = {"cfg3", "cfg4", "closed-loop", "backbone", "corr-sweep"}
    skipped = []

    def leg(name):
        """True when the optional leg `name` was asked for and the wall-time budget still allows STARTING it."""
        if name not in extras:
            return False
        if args.budget_s > 0 and time.perf_counter() - t_start > args.budget_s:
            skipped.append(name + ":budget")
            return False
        return True

    def phase(name, t0):
        phases[name] = round(phases.get(name, 0.0) + time.perf_counter() - t0, 2)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # one process per GPU over RCCL: re-launch under torch.distributed.run
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or let `python bench.py --gpus N` launch them)")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    # AOC_DIST_BACKEND=gloo (developer switch): several ranks on ONE GPU with host-side reductions, to exercise the multi-rank code path
    # where no multi-GPU node is at hand; the default is RCCL ("nccl"), one rank per GPU
    backend = os.environ.get("AOC_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    red_dev = dev if backend == "nccl" else None          # where the few reduced scalars live
    # AOC_DIST_FORCE=1 (test switch): initialise the process group even with ONE rank, so that every barrier / all-reduce of this file goes through RCCL
    # on a one-GPU box (tests/test_gpu_eval_loop.py: the first 8-GPU run must not be the first time RCCL executes)
    use_dist = world > 1 or os.environ.get("AOC_DIST_FORCE", "") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    if args.cu_reserve >= n_cu:
        args.cu_reserve = 0
    aoc_amd._lib.lib()
    if args.cu_reserve > 0:
        aoc_amd.ops.set_stream_cus(n_cu - args.cu_reserve)   # the matrix kernels size their grids in whole rounds of the CUs their stream may use

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            torch.distributed.barrier()

    if args.eval_sharded:
        # ---- BASELINE.json configs[4]: a fixed synthetic sequence set partitioned over the ranks (LPT), one all-reduce at the end
        specs = eval_runner.make_sequence_set("cfg5", scale=args.eval_scale, seed=0)
        np.random.seed(1234 + rank)
        with torch.no_grad():
            eval_runner.eval_sharded(specs[:1], 0, 1, dev, max_frames=3)          # warm-up: allocator, library load
            # the rank's sequences are synthesised and made resident first; the timed region starts at the barrier inside
            tot = eval_runner.eval_sharded(specs, rank, world, dev, barrier=barrier, lanes=max(1, args.eval_lanes))
            barrier()
        el = torch.tensor([tot["loop_seconds_max"]], dtype=torch.float64)
        if rank == 0:
            line = {"metric": "frames/sec, AOC-Net matching + read-out + memory policy, sequence-sharded evaluation", "value": round(tot["frames"] / float(el.item()), 3),
                    "unit": "frames/s", "n_gpus": world, "steps": int(tot["frames"]), "warmup": 3, "ms_per_step": round(float(el.item()) / max(tot["frames"], 1) * 1e3, 4),
                    "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": f"cfg5: {len(specs)} synthetic sequences ({args.eval_scale:g} of 30 DAVIS-17-val-like 121x213 K=16 + 507 YouTube-VOS-19-like "
                                           "145x261 K in {8,16,32}), closed loop (matching -> DynamicPreHead -> linear read-out -> soft-max -> memory policy), "
                                           "reference-API path (one host read-back of the row counts per frame for scipy's initial rows); "
                                           f"{max(1, args.eval_lanes)} sequences in flight per rank on separate HIP streams",
                               "sharding": "sequences over ranks by LPT on frames x objects, no data-path collective; one all-reduce(SUM) + one all-reduce(MAX) of metric accumulators"},
                    "eval": {k: tot[k] for k in ("sequences", "ranks", "frames", "objects", "mean_j", "mean_f", "rank_seconds_max", "rank_seconds_mean",
                                                 "imbalance", "planned_imbalance")},
                    "roofline": None, "cpu_baseline": None, "ranks_seen": int(tot["ranks"]), "imbalance": round(float(tot["imbalance"]), 4), "schema": SCHEMA}
            report = json.dumps(line)
            assert len(report) < COMPACT_LIMIT, len(report)
        finish(report if rank == 0 else None, use_dist)
        return

    cfg = syn.CONFIGS[args.config]
    levels = CONFIG_LEVELS.get(args.config)
    mc = hotpath.MatchingConfig(CLUSTER_NUM=cfg.k, CLUSTER_LEVELS=levels)
    torch.manual_seed(0)
    gates = hotpath.CalibrationGates(mc).to(dev)
    n_streams = max(1, args.streams)
    # sequences are sharded over ranks: rank r owns sequences r*n_streams .. (+n_streams)
    chain_plan = args.chain_plan if args.chain_plan is not None else ("2,3" if args.config == "cfg2" else "")
    if chain_plan == "none":
        chain_plan = ""
    workloads = [ClipWorkload(cfg, seed=1 + rank * n_streams + s, device=dev, mc=mc, overlap=not args.no_overlap,
                              phase=s, start=(s * mc.MEM_EVERY) // n_streams if args.stagger else 0) for s in range(n_streams)]
    for wl in workloads:
        wl.chains = 1 if args.reuse_proxies else max(1, min(args.chains, mc.MEM_EVERY))
        wl.chain_plan = [int(x) for x in chain_plan.split(",")] if (chain_plan and not args.reuse_proxies) else None
        wl.chain_lead = max(1, args.chain_lead)
        if wl.side is not None and args.chain_streams > 1:
            wl.sides = [wl.side] + [torch.cuda.Stream(device=dev, priority=-1) for _ in range(args.chain_streams - 1)]
        wl.reuse_proxies = args.reuse_proxies
        if args.incremental_proxies:
            wl.enable_incremental()
        if not args.python_frames and not args.incremental_proxies and hotpath.FrameRunner.supported(mc, cfg.c, cfg.n_obj):
            wl.runner = hotpath.FrameRunner(mc, cfg.h, cfg.w, cfg.c, cfg.n_obj, wl.rmax, dev)
        wl.batch_gates = not args.python_frames
    acts = make_activations(gates, cfg.n_obj, cfg.h, cfg.w, dev, seed=7)

    def make_main_stream():
        """Main stream of one sequence.  With --cu-reserve N its workgroups are kept off N of the CUs (HIP CU mask), so
        that the latency-bound k-means chain on the side stream always finds free CUs next to the dense kernel, whose
        blocks fill a CU's register file and do not yield."""
        if args.cu_reserve <= 0:
            return torch.cuda.Stream(device=dev)
        hip = ctypes.CDLL("libamdhip64.so")
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
            aoc_amd.ops.set_stream_cus(0)
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
    batch_corr = args.batch_corr and n_streams > 1 and cfg.c == 100

    hw, C, O = cfg.h * cfg.w, cfg.c, cfg.n_obj

    def meta_dense(query_flat, pool, prep, *a, **k):
        m, n = query_flat.shape[0], pool.shape[0]      # synthetic labels: every pool pixel is kept
        return dict(flops=2.0 * m * n * C, bytes=(m + n) * C * 4 + n * 4 + m * O * 4)

    def meta_dense_split(query_flat, query_split, pool, pool_split, prep, *a, **k):
        return meta_dense(query_flat, pool[:prep.n], prep)

    def meta_proxy(query_flat, proxies, proxy_sqnorm, set_begin, *a, **k):
        m, npx = query_flat.shape[0], proxies.shape[0]
        return dict(flops=2.0 * m * npx * C, bytes=m * C * 4 + npx * C * 4 + m * len(set_begin) * 4)

    def meta_proxy_batched(frames, set_begin, *a, **k):
        m, npx = frames[0][0].shape[0], frames[0][1].shape[0]
        return dict(flops=2.0 * m * npx * C * len(frames), bytes=(m * C * 4 + npx * C * 4 + m * len(set_begin) * 4) * len(frames), frames=len(frames))

    def meta_proxy_records(frames, set_begin, *a, **k):
        m, npx = frames[0][0].shape[0], frames[0][2].shape[0]
        return dict(flops=2.0 * m * npx * C * len(frames), bytes=(m * C * 4 + npx * C * 4 + m * len(set_begin) * 4) * len(frames), frames=len(frames))

    def meta_kmeans(pool, rows, seg_offsets, seg_k, init_rows, kmax, iters=20, rows_capacity=None, n_rep=1):
        n = pool.shape[0]
        return dict(flops=2.0 * iters * n * kmax * C, bytes=float(iters) * n * C * 4 * 2 + n * 4)

    def meta_chain(pool, prep, levels, init_rows, tables, sqnorms, iters=20):
        n = pool.shape[0]
        return dict(flops=2.0 * iters * n * max(levels) * C, bytes=float(iters) * n * C * 4 * 2 + n * 4)

    def meta_film(x, *a, **k):
        return dict(flops=float(x.numel()), bytes=2.0 * x.numel() * 4)          # 2 O c h w s

    def meta_cond(z, *a, **k):
        return dict(flops=2.0 * z.numel(), bytes=1.0 * z.numel() * 4)           # O C H W s (one read of z)

    # HIP-event pairs only around the ops the roofline objects need (an event pair costs ~25 us of host time, and the host
    # enqueues ~60 ops per frame); every kernel's duration is in the rocprofv3 summary under profiles/
    timer = OpTimer(["dense_match_min", "dense_match_min_split", "proxy_corr_min", "proxy_corr_min_batched", "proxy_corr_min_records", "kmeans_segmented", "cluster_chain",
                     "local_window_match",
                     "film_scale", "cond_gate_pool"])
    def frame_meta(kind, m, n):
        if kind == "dense":
            return dict(flops=2.0 * m * n * C, bytes=(m + n) * C * 4 + n * 4 + m * O * 4)
        n_set = 2 * len(mc.cluster_levels) * O + O
        return dict(flops=2.0 * m * n * C, bytes=m * C * 4 + n * C * 4 + m * n_set * 4, frames=1)
    timer.frame_meta = frame_meta
    for wl in workloads:
        wl.timer = timer
    timer.serialize_dense = bool(args.dense_order) and not args.no_dense_order
    timer.install(dict(dense_match_min=meta_dense, dense_match_min_split=meta_dense_split, proxy_corr_min=meta_proxy,
                       proxy_corr_min_batched=meta_proxy_batched, proxy_corr_min_records=meta_proxy_records, kmeans_segmented=meta_kmeans, cluster_chain=meta_chain, film_scale=meta_film, cond_gate_pool=meta_cond))
    corr_stream = torch.cuda.Stream(device=dev) if batch_corr else None

    def run_steps(n):
        for _ in range(n):
            pend = []
            for wl, st in zip(workloads, streams):
                if n_streams > 1 or (args.cu_reserve > 0 and args.mask_main):
                    with torch.cuda.stream(st):
                        _, _, p = frame_step(wl, gates, acts, args.dense, not args.no_pipeline, batch_corr)
                else:
                    _, _, p = frame_step(wl, gates, acts, args.dense, not args.no_pipeline, batch_corr)
                pend.append(p)
            if batch_corr:
                # ONE correlation launch for the frames of all in-flight sequences; every sequence's stream joins it
                done = hotpath.launch_correlations(pend, corr_stream)
                for st in streams:
                    st.wait_event(done)

    with torch.no_grad():
        t_ph = time.perf_counter()
        for wl, st in zip(workloads, streams):             # allocator pre-touch at every pool size (setup)
            with torch.cuda.stream(st):
                wl.pretouch(gates, acts, args.dense, not args.no_pipeline)
        run_steps(args.warmup)
        barrier()
        phase("setup_and_warmup", t_start)
        t_ph = time.perf_counter()
        ops.dense_prune_stats(reset=True)     # counters of the coarse-then-rescore dense kernel over the timed regions (reads synchronise: outside them)
        for wl in workloads:
            wl.count_r = True
        timer.want_segments = args.segments
        if args.segments:
            args.probe_every = 1                  # consecutive frames of a stream: "to_next_frame" is then the gap to the very next frame
        timer.probe_every = args.probe_every
        for i, wl in enumerate(workloads):
            wl.probe_phase = i * max(1, args.probe_every // 2)
        timer.enabled = True                 # HIP events around every op, on the stream the op is launched on
        # EXACTLY --steps steps per region, bracketed by barrier + synchronize; regions are repeated (the group walk simply continues) until
        # --min-region-s seconds are covered, and the line reports the median region
        regions, enqueue = [], []
        while True:
            t0 = time.perf_counter()
            run_steps(args.steps)
            enqueue.append(time.perf_counter() - t0)
            barrier()
            regions.append(time.perf_counter() - t0)
            more = torch.tensor([1.0 if (sum(regions) < args.min_region_s and len(regions) < 15) else 0.0], dtype=torch.float64, device=red_dev)
            if use_dist:
                torch.distributed.all_reduce(more, op=torch.distributed.ReduceOp.MAX)      # every rank runs the same number of regions
            if float(more.item()) == 0.0:
                break
        timer.enabled = False
        for wl in workloads:
            wl.count_r = False
        prune = ops.dense_prune_stats(reset=True)            # exactly the timed regions (the drained / profiled steps below are not counted)
        phase("timed_regions", t_ph)
        t_ph = time.perf_counter()
        # What the enqueue calls of a step cost the host when nothing blocks: a few extra steps (outside the timed regions, the group walk simply
        # continues) with the queues drained in front of each -- the wall time of run_steps(1) is then the CPU time of the step's enqueue calls,
        # while `enqueue` above also contains the time the host sits in the runtime waiting for room in full queues (it runs ~3 ms ahead)
        cpu_enqueue = []
        timer.want_segments = False               # (the drained steps below are not part of any reported bracket)
        n_drained = max(8, min(40, syn.CONFIGS[args.config].frames - 1))       # one walk over the clip: steps with and without chain launches in proportion
        for _ in range(n_drained):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_steps(1)
            cpu_enqueue.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        if args.host_profile and rank == 0:
            import cProfile
            import pstats
            prof = cProfile.Profile()
            for _ in range(40):
                torch.cuda.synchronize()
                prof.enable()
                run_steps(1)
                prof.disable()
            torch.cuda.synchronize()
            with open(args.host_profile, "w") as f:
                pstats.Stats(prof, stream=f).sort_stats("cumulative").print_stats(45)
                pstats.Stats(prof, stream=f).sort_stats("tottime").print_stats(30)
        host_cpu_enqueue_s = sum(cpu_enqueue) / len(cpu_enqueue)
        phase("drained_steps", t_ph)
    n_regions = len(regions)

    el = torch.tensor(regions, dtype=torch.float64, device=red_dev)
    if use_dist:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)                  # per region: the slowest rank
    region_max = [float(x) for x in el.tolist()]
    med = sorted(range(n_regions), key=lambda i: region_max[i])[n_regions // 2]
    elapsed, host_enqueue_s = regions[med], enqueue[med]
    elapsed_max = region_max[med]
    frames_local = args.steps * n_streams
    metrics = sharding.allreduce_metrics(dict(frames=frames_local, objects=frames_local * (O - 1), gpu_seconds=elapsed), device=red_dev)
    # what every rank contributed (one more tiny exchange, outside the timed regions): frames and seconds of the median region per rank
    per_rank = torch.zeros(world, 2, dtype=torch.float64, device=red_dev)
    per_rank[rank, 0], per_rank[rank, 1] = frames_local, elapsed
    if use_dist:
        torch.distributed.all_reduce(per_rank, op=torch.distributed.ReduceOp.SUM)
    per_rank = per_rank.cpu().tolist()
    ranks_seen = torch.distributed.get_world_size() if use_dist else 1
    r_hist = {}
    for wl in workloads:
        for r, c in wl.r_hist.items():
            r_hist[r] = r_hist.get(r, 0) + c

    probe_ms = timer.kernel_probe.elapsed_ms()
    segments = timer.segment_summary() if args.segments else None
    if segments and HOST_S["frames"]:
        segments["host_ms_per_frame (chain launch, frame call, gates call; warm-up included)"] = [round(HOST_S[k] / HOST_S["frames"] * 1e3, 4)
                                                                                               for k in ("chain_launch", "frame_call", "gates_call")]
    if args.dump_timeline and rank == 0:
        with open(args.dump_timeline, "w") as f:
            json.dump(timer.timeline(), f)

    # second, informational region (N = 1 only): a few steps with the exact-fp32 dense kernel (`--dense fp32`), so that the line also
    # carries the figure of the all-fp32 arithmetic next to the headline
    exact = None
    if world == 1 and args.dense == "split" and args.exact_steps > 0:
        t_ph = time.perf_counter()
        with torch.no_grad():
            for wl in workloads:
                wl.reset()
            saved = args.dense
            args.dense = "fp32"
            run_steps(2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(args.exact_steps)
            torch.cuda.synchronize()
            e2 = time.perf_counter() - t1
            args.dense = saved
        exact = dict(value=round(args.exact_steps * n_streams / e2, 3), unit="frames/s", steps=args.exact_steps, ms_per_step=round(e2 / args.exact_steps * 1e3, 4),
                     note="same workload with aoc_dense_match_min (v_mfma_f32_16x16x4_f32) instead of the fp16-split kernel; a shorter region that starts "
                          "at the beginning of the group walk")
        phase("exact_fp32_region", t_ph)
    # ---- optional leg `closed-loop` (BASELINE.json configs[4]): a FIXED synthetic sequence set partitioned over the ranks (LPT on frames x objects),
    # the closed evaluation loop per rank, one all-reduce(SUM) + one all-reduce(MAX) at the end
    strong = None
    run_closed = leg("closed-loop") and args.config == "cfg2" and not (args.reuse_proxies or args.incremental_proxies)
    if use_dist:                                          # every rank takes the same decision (the budget is a wall clock)
        flag = torch.tensor([1.0 if run_closed else 0.0], dtype=torch.float64, device=red_dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        run_closed = float(flag.item()) > 0
    if run_closed:
        t_ph = time.perf_counter()
        specs = eval_runner.make_sequence_set("cfg5", scale=args.strong_scale, seed=0)
        np.random.seed(1234 + rank)
        with torch.no_grad():
            eval_runner.eval_sharded(specs[:1], 0, 1, dev, max_frames=3)          # warm-up of this path's allocations
            first = eval_runner.eval_sharded(specs, rank, world, dev, barrier=barrier, lanes=max(1, args.eval_lanes))
            barrier()
            tot = eval_runner.eval_sharded(specs, rank, world, dev, barrier=barrier, lanes=max(1, args.eval_lanes))       # the same set once more
            barrier()
        secs_first, secs = float(first["loop_seconds_max"]), float(tot["loop_seconds_max"])
        # like for like with `value`: the same closed loop over DAVIS-17-like sequences only (121x213 maps, K = 16, 1-3 objects: the cfg2 shape; the
        # mixed set above is 95 % YouTube-VOS-like sequences with three cluster levels, whose orchestrated figure is other_configs.cfg3)
        dspecs = eval_runner.make_sequence_set("davis17", scale=max(args.strong_scale, 0.27), seed=0)
        with torch.no_grad():
            dfirst = eval_runner.eval_sharded(dspecs, rank, world, dev, barrier=barrier, lanes=max(1, args.eval_lanes))
            barrier()
            dtot = eval_runner.eval_sharded(dspecs, rank, world, dev, barrier=barrier, lanes=max(1, args.eval_lanes))      # the same sequences once more
            barrier()
        # `value` keeps its rounds-1-4 meaning (ONE pass over the set, the first); the warm second pass has its own key
        davis = dict(value=round(dfirst["frames"] / float(dfirst["loop_seconds_max"]), 3), unit="frames/s", sequences=int(dtot["sequences"]), frames=int(dtot["frames"]),
                     mean_objects_incl_background=round(float(np.mean([sp.n_obj for sp in dspecs])), 2),
                     second_pass_value=round(dtot["frames"] / float(dtot["loop_seconds_max"]), 3))
        strong = dict(metric="frames/sec, sequence-sharded evaluation of a fixed set (strong scaling)", value=round(first["frames"] / secs_first, 3), unit="frames/s",
                      second_pass_value=round(tot["frames"] / secs, 3),
                      n_gpus=world, sequences=int(tot["sequences"]), frames=int(tot["frames"]), objects=int(tot["objects"]),
                      loop_seconds_max=round(secs_first, 4), rank_seconds_mean=round(float(first["rank_seconds_mean"]), 4),
                      imbalance=round(float(first["imbalance"]), 4), planned_imbalance=round(float(first["planned_imbalance"]), 4),
                      mean_j=tot["mean_j"], mean_f=tot["mean_f"], lanes_per_rank=max(1, args.eval_lanes), closed_loop_davis17_like=davis,
                      workload=f"{len(specs)} synthetic sequences = {args.strong_scale:g} of the 30 DAVIS-17-val-like (121x213, K=16) + 507 YouTube-VOS-19-like "
                               "(145x261, K in {8,16,32}) set; closed loop (matching -> DynamicPreHead -> linear read-out -> soft-max -> memory policy) through "
                               "the reference-API path; the set does not depend on the number of ranks")
        phase("closed_loop", t_ph)
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
            # the single kernel with the largest share of GPU time (profiles/: dense_prune_kernel); the k-means op is a chain of
            # ~85 small launches per call and is reported as its own object below
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
                # algorithmic flops (2 m n C) against the fp16 pipe the kernel runs on; the instruction stream executes ONE
                # fp16 product (K padded 100 -> 112) for every (reference tile, query tile) pair and the two cross products only for the
                # pairs that could hold a maximum (counted by the kernel over the timed regions)
                rescored = prune["rescored"] / max(prune["tested"], 1)
                executed = k["tflops"] * (1.0 + 2.0 * rescored) * 112.0 / C
                traffic_file = "profiles/r06_pmc_dense_R6.txt"
                roofline = dict(kernel="dense_prune_kernel (aoc_dense_match_min_split), an fp16-pipe kernel: algorithmic fp32 flops priced against the DENSE FP16 MFMA peak",
                                bound="mfma", achieved=k["tflops"],
                                peak=PEAK_F16_MFMA_TFLOPS, unit="TFLOP/s", frac=round(k["tflops"] / PEAK_F16_MFMA_TFLOPS, 4),
                                traffic=pmc_traffic_bytes(traffic_file) if args.config == "cfg2" else None,
                                traffic_source=f"{traffic_file}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the kernel alone at R = 6 (FETCH x 2 + WRITE), "
                                               "read from the committed file at bench time -- counters cannot be collected inside the timed run",
                                avg_launch_ms=k["avg_ms"], algorithmic_flops_per_launch=k["avg_flops"],
                                executed_tflops=round(executed, 1), pipe_frac=round(executed / PEAK_F16_MFMA_TFLOPS, 4),
                                rescored_pair_fraction=round(rescored, 4),
                                note="fp32-equivalent distances from fp16 MFMAs: one hi*hi product prunes, the pairs that survive get hi*lo + lo*hi "
                                     "added on chip (exactly the three-product value; same result as evaluating everything); frac prices the "
                                     "ALGORITHMIC fp32 flops (2 m n C) against the dense fp16 peak, pipe_frac the executed ones; avg_launch_ms = "
                                     "hipEvents recorded by the library immediately around the kernel while the other sequence's stream shares the GPU",
                                op_avg_ms=k.get("op_avg_ms"))
                if args.cu_reserve > 0:
                    # the kernel is launched on a stream whose CU mask leaves cu_reserve CUs to the k-means chains
                    cus = n_cu - args.cu_reserve
                    roofline["cus_available_to_kernel"] = cus
                    roofline["pipe_frac_of_available_cus"] = round(executed / (PEAK_F16_MFMA_TFLOPS * cus / n_cu), 4)
        km = kernels.get("cluster_chain") or kernels.get("kmeans_segmented")
        km_roof = None
        if km and "gbs" in km:
            km_roof = dict(kernel="aoc_cluster_chain_enqueue (20 Lloyd iterations of the 1 to 3 frames that share a chain + proxy construction)", bound="hbm",
                           achieved=km["gbs"], peak=PEAK_HBM_GBS, unit="GB/s", frac=round(km["gbs"] / PEAK_HBM_GBS, 4), traffic=None,
                           avg_launch_ms=km["avg_ms"], algorithmic_bytes_per_launch=km["avg_bytes"],
                           offline={"traffic": "profiles/r06_pmc_kmeans_R6_F3.txt", "alone": "profiles/r06_kmeans_chain_events.txt"},
                           note="a dependent chain of ~85 launches whose ordered float32 sums are latency-bound by construction; the in-run "
                                "figure spans the time the chain shares the GPU with the other streams")

        def hbm_roof(name, kernel, note, offline):
            kk = kernels.get(name)
            if not kk or "gbs" not in kk:
                return None
            return dict(kernel=kernel, bound="hbm", achieved=kk["gbs"], peak=PEAK_HBM_GBS, unit="GB/s", frac=round(kk["gbs"] / PEAK_HBM_GBS, 4), traffic=None,
                        avg_launch_ms=kk["avg_ms"], algorithmic_bytes_per_launch=kk["avg_bytes"], calls=kk["calls"], note=note, offline=offline)

        film_roof = hbm_roof("film_scale", "film_scale_ahead_kernel (aoc_film_scale: IA gates and the FiLM of the conditioning blocks)",
                             "in-run average over the 14 activation shapes of decoding_module.py:22-84; algorithmic bytes 2 O c h w 4 (SURVEY 8d)",
                             {"traffic": "profiles/r04_pmc_gates_cfg2.txt", "alone": "profiles/r06_gates_standalone.txt"})
        cond_roof = hbm_roof("cond_gate_pool", "cond_scores_part / cond_scores_reduce / cond_select_tail / cond_masked_gap_fused (aoc_cond_gate_pool_ex)",
                             "in-run average over the 4 conditioning blocks; algorithmic bytes O C H W 4 = ONE read of z (SURVEY 8d); the op reads z twice "
                             "(scores, masked pooling) around the exact k-th-largest selection",
                             {"traffic": "profiles/r04_pmc_gates_cfg2.txt", "second_read": "profiles/r05_cond_second_read.txt"})
        corr_name = next((k for k in ("proxy_corr_min_records", "proxy_corr_min_batched") if k in kernels), "proxy_corr_min")
        corr = kernels.get(corr_name)
        corr_roof = None
        if corr and "gbs" in corr:
            # north_star grades this kernel against the HBM roofline.  `frac` is the KERNEL's own figure at the product's frames per launch: an event pair
            # around the launch on an otherwise idle GPU; the in-run bracket, which measures the schedule as much as the kernel, is `in_run_frac`
            corr_roof = dict(kernel={"proxy_corr_min_records": "proxy_corr_records_kernel (aoc_proxy_corr_min_records: the query as the tile-major "
                                                               "split records the dense kernel consumes)",
                                     "proxy_corr_min_batched": "proxy_corr_batched_kernel (aoc_proxy_corr_min_batched)"}.get(corr_name, "proxy_corr_min_kernel"),
                             bound="hbm", achieved=corr["gbs"], peak=PEAK_HBM_GBS, unit="GB/s",
                             frac=round(corr["gbs"] / PEAK_HBM_GBS, 4), traffic=None, avg_launch_ms=corr["avg_ms"], algorithmic_bytes_per_launch=corr["avg_bytes"],
                             frames_per_launch=n_streams if batch_corr else 1, in_run_frac=round(corr["gbs"] / PEAK_HBM_GBS, 4),
                             offline={"traffic_16_frames_per_launch": "profiles/r03_pmc_corr_records_B16.txt"})
            if C == 100:
                t_ph = time.perf_counter()
                with torch.no_grad():
                    iso = correlation_roofline(cfg, mc, dev, batches=(1, 4, 16, 32) if leg("corr-sweep") else (1, 16), reps=20)
                corr_roof["isolated"] = iso
                own = next((r for r in iso if r["frames_per_launch"] == corr_roof["frames_per_launch"]), None)
                best = max(iso, key=lambda r: r["frac"])
                corr_roof.update(best_frac=best["frac"], best_frames_per_launch=best["frames_per_launch"])
                if own is not None:
                    corr_roof.update(achieved=own["achieved"], frac=own["frac"], avg_launch_ms=own["avg_launch_ms"],
                                     frac_including_query_split=own["frac_including_query_split"])
                phase("correlation_alone", t_ph)

        cpu = None
        parity = None
        if world == 1 and not args.no_cpu_baseline:
            t_ph = time.perf_counter()
            acts_cpu = [a.cpu() for a in acts]
            feats, rows, (emb2, lab2), tm, t_match, t_cal, threads, frame_s = cpu_baseline(cfg, mc, 1, gates, acts_cpu, runs=args.cpu_runs)
            br = ", ".join(f"{k} {v:.2f}" for k, v in tm.items())
            cpu = dict(value=round(1.0 / (t_match + t_cal), 5), unit="frames/s", cores=threads, kind="port", runs=len(frame_s),
                       sample=f"median of {len(frame_s)} runs after 1 warm-up: 1 {cfg.name} frame, R={CPU_BASELINE_R} pool frames, "
                              f"dense branch on {'every' if DENSE_SUBSAMPLE == 1 else f'every {DENSE_SUBSAMPLE}th'} query pixel"
                              f"{'' if DENSE_SUBSAMPLE == 1 else f' x{DENSE_SUBSAMPLE}'}; matching {t_match:.2f} s + gates {t_cal:.2f} s",
                       frame_seconds=frame_s, min_frame_seconds=min(frame_s), value_of_min=round(1.0 / min(frame_s), 5),
                       branches_seconds_of_median_run={k: round(v, 4) for k, v in tm.items()},
                       detail=f"matching [{br}] + calibration gates {t_cal:.2f} s (each distinct gate shape timed once per run x its count); torch CPU fp32 + C k-means "
                              f"oracle, {threads} threads; the clip's average pool is 6.4 frames")
            phase("cpu_baseline", t_ph)
            t_ph = time.perf_counter()
            # parity spot check of a frame (R = 1) on the GPU: every branch against the oracle, and a surrogate
            # mask (argmin over objects of the dense-matching channel) on the sub-sampled pixels
            e2, l2 = emb2.to(dev), lab2.to(dev)
            with torch.no_grad():
                fg, _, _ = hotpath.proto_mask_features(mc, e2[:1], l2[:1], e2[0], l2[0], e2[1], workloads[0].bias, init_rows=rows)
            fg = fg.cpu()
            chs = hotpath.channel_slices(mc)
            nl = len(mc.MODEL_MULTI_LOCAL_DISTANCE)
            ncl = 2 * len(mc.cluster_levels)
            diffs = {
                "dense": float((fg[:, 0].reshape(O, -1)[:, ::DENSE_SUBSAMPLE].t() - feats["dense_sub"]).abs().max()),
                "cluster": float((fg[:, chs["cluster"]:chs["cluster"] + ncl] - feats["cluster"][0].permute(2, 3, 0, 1)).abs().max()),
                "proxy": float((fg[:, chs["proxy"]:chs["proxy"] + 1] - feats["proxy"][0].permute(2, 3, 0, 1)).abs().max()),
                "local": float((fg[:, chs["local"]:chs["local"] + nl] - feats["local"][0].permute(2, 3, 0, 1)).abs().max()),
                "local_proxy": float((fg[:, chs["local_proxy"]:chs["local_proxy"] + nl] - feats["local_proxy"][0].permute(2, 3, 0, 1)).abs().max()),
            }
            pg = fg[:, 0].reshape(O, -1)[:, ::DENSE_SUBSAMPLE].argmin(0)
            pc = feats["dense_sub"].argmin(1)
            iou_sum, iou_n = sharding.mask_iou_sums(pg, pc, O)
            parity = dict(max_abs_feature_diff=max(diffs.values()), per_branch=diffs, surrogate_mask_mean_iou=iou_sum / iou_n)
            phase("parity", t_ph)

        other = {}
        if world == 1 and args.config == "cfg2":
            for name in ("cfg3", "cfg4"):
                if not leg(name):
                    continue
                t_ph = time.perf_counter()
                try:
                    # a region = whole walks over the config's clip (frames - 1 steps each), like the cfg2 default: every pool size in proportion
                    walk = syn.CONFIGS[name].frames - 1
                    child_details = os.path.splitext(args.details_file)[0] + f"_{name}.json"
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", name, "--steps", str(walk * (2 if walk < 10 else 1)), "--warmup", "3",
                                        "--min-region-s", "0.3", "--no-cpu-baseline", "--exact-steps", "0", "--no-extras", "--details-file", child_details],
                                       capture_output=True, text=True, timeout=300)
                    j = json.loads(r.stdout.strip().splitlines()[-1])
                    other[name] = dict(value=j["value"], unit=j["unit"], ms_per_step=j["ms_per_step"], steps=j["steps"], timed_regions=j["timed_regions"],
                                       workload=j["config"]["workload"], host_enqueue_ms_per_step=j["host_enqueue_ms_per_step"],
                                       host_enqueue_cpu_ms_per_step=j.get("host_enqueue_cpu_ms_per_step"), details_file=j.get("details_file"),
                                       roofline=j.get("roofline"), roofline_correlation=j.get("roofline_correlation"),
                                       roofline_kmeans_chain=j.get("roofline_kmeans_chain"), roofline_film_scale=j.get("roofline_film_scale"))
                except Exception as e:                      # the headline must not depend on the extras
                    other[name] = dict(error=repr(e)[:200])
                phase(name, t_ph)
        # SURVEY 8(d) "reported separately": plain-PyTorch random-weight ResNet101-DeepLabv3+ in front of the hot path (tools/backbone_e2e.py; the
        # backbone is out of scope -- this only says what share of an image-level frame the hot path is)
        e2e = None
        if world == 1 and args.config == "cfg2" and leg("backbone"):
            t_ph = time.perf_counter()
            try:
                r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "backbone_e2e.py")],
                                   capture_output=True, text=True, timeout=240)
                e2e = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:
                e2e = dict(error=repr(e)[:200])
            phase("backbone", t_ph)
        value = metrics["frames"] / elapsed_max
        rs = sorted(r_hist)
        rank_seconds = [p[1] for p in per_rank]
        full = {
            "metric": "frames/sec, AOC-Net matching + calibration hot path (480p, 3 objects)",
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "schema": SCHEMA,
            "dtype_detail": ("results fp32-equivalent (max 1.5e-6 on the proto-mask features against the exact-fp32 kernels and the CPU oracle): dense matching and "
                             "the correlation kernel multiply fp16-split operands (x * 2^10 = hi + lo; hi*hi + hi*lo + lo*hi) on the fp16 matrix pipe with fp32 "
                             "accumulate, exact-fp32 take-over on the device when a precondition fails; k-means (bit-exact to scipy), local matching and the "
                             "calibration gates are fp32 throughout; exact_fp32_value = the same workload with the fp32-MFMA dense kernel"
                             if args.dense == "split" else "fp32 throughout"),
            "timed_regions": n_regions, "region_ms": [round(x * 1e3, 3) for x in region_max],
            "ranks_seen": ranks_seen, "frames_per_rank": [int(p[0]) for p in per_rank],
            "imbalance": round(max(rank_seconds) / (sum(rank_seconds) / len(rank_seconds)), 4),
            "config": {"workload": (f"{cfg.name}: {cfg.h}x{cfg.w} stride-4 maps{' (480p)' if (cfg.h, cfg.w) == (121, 213) else ''}, O={O} ({O - 1} objects + background), "
                                    f"K={'/'.join(str(k) for k in mc.cluster_levels)} proxies, C={C}, {cfg.frames}-frame clips, MEM_EVERY={mc.MEM_EVERY} "
                                    f"(pool R=1..{workloads[0].rmax}), 20 Lloyd iterations, local windows 2..12, 10 IA gates + 4 conditioning blocks"),
                       "R_mean": round(sum(r * c for r, c in r_hist.items()) / max(sum(r_hist.values()), 1), 3),
                       "R_timed": {"histogram": {str(r): r_hist[r] for r in rs},
                                   "clip_mean": round(float(np.mean([workloads[0].R_of(t) for t in range(1, cfg.frames)])), 3)},
                       "sequences_per_gpu": n_streams, "frames_per_step": n_streams, "sharding": "sequences over ranks, no data-path collective",
                       "intra_frame_overlap": ("none" if args.no_overlap else "k-means chain on a side HIP stream" +
                                               ("" if args.no_pipeline else ", enqueued as soon as the pool it depends on is final")),
                       "kmeans_chain_batches": (f"the k-means of the {mc.MEM_EVERY} frames that see one pool state advance as chains of {chain_plan} frames" if chain_plan
                                                else f"the frame that needs it first alone, the next {max(0, args.chains - 1)} batched into one chain"),
                       "correlation": ("ONE aoc_proxy_corr_min_records launch per step for the frames of all in-flight sequences" if batch_corr
                                       else "one aoc_proxy_corr_min_records launch (fp16-split kernel on the query's split records, one frame) per sequence and frame"),
                       "cu_reserve": (f"main streams masked off {args.cu_reserve} of {n_cu} CUs (hipExtStreamCreateWithCUMask), left to the side-stream "
                                      "k-means chains" if args.cu_reserve > 0 else "none"),
                       "proxy_mode": ("NON-PARITY incremental" if args.incremental_proxies else "NON-PARITY reuse" if args.reuse_proxies else "reference"),
                       "proxy_mode_detail": ("NON-PARITY: every reference frame clustered once when it joins the pool, frames matched against the union of the "
                                             "per-frame code books (hotpath.IncrementalProxyBank)" if args.incremental_proxies else
                                             "NON-PARITY: adaptive proxies reused until the pool changes (one k-means per MEM_EVERY frames)"
                                             if args.reuse_proxies else "reference: the pool is re-clustered for every frame with that frame's initial rows"),
                       "frame_call": ("ONE aoc_frame_enqueue call per frame (the counterpart of the single before_seghead_process call, aocnet.py:114)"
                                      if workloads[0].runner is not None else "hotpath.proto_mask_features: the individual C entry points, ~45 ctypes calls per frame"),
                       "dense_precision": ("fp16-split products (hi*hi + hi*lo + lo*hi), fp32 accumulate: fp32-equivalent; exact-fp32 take-over "
                                           "on overflow / soft labels" if args.dense == "split" else "exact fp32 MFMA")},
            # `host_enqueue_ms_per_step`: wall time of the timed region's enqueue loop per step (the meaning of rounds 1-4; it also counts the time the host
            # waits for room in full queues).  `host_enqueue_cpu_ms_per_step`: the CPU time of one step's enqueue calls, queues drained in front of the step
            # (mean over one walk of the clip, at most 40 steps; round 5 printed this under the first key)
            "host_enqueue_ms_per_step": round(host_enqueue_s / args.steps * 1e3, 3),
            "host_enqueue_cpu_ms_per_step": round(host_cpu_enqueue_s * 1e3, 3),
            "probe_sampling": (f"the HIP-event brackets behind `kernels` / the roofline objects are placed on every {args.probe_every}-th frame of each sequence "
                               "inside the timed region (on every frame -- --probe-every 1 -- their ~40 event records per frame cost 2.6 % frames/s)"),
            **({"frame_segments_ms": segments} if segments else {}),
            "exact_fp32_dense_run": exact, "exact_fp32_value": exact["value"] if exact else None,
            "cfg3_value": other.get("cfg3", {}).get("value"), "cfg4_value": other.get("cfg4", {}).get("value"),
            "closed_loop_value": strong["value"] if strong else None,
            # `roofline` = the dominant kernel (largest share of GPU time: profiles/r06_kernel_stats_cfg2.csv); `roofline_correlation` = the kernel north_star grades
            "roofline": roofline, "roofline_correlation": corr_roof,
            "roofline_kmeans_chain": km_roof, "roofline_film_scale": film_roof,
            "roofline_cond_gate_pool": cond_roof, "strong_scaling": strong, "other_configs": other or None, "image_level_end_to_end": e2e, "cpu_baseline": cpu, "parity": parity,
            "kernels": kernels, "skipped": skipped or None, "phases_s": phases, "argv": sys.argv[1:],
        }
        full["wall_s"] = round(time.perf_counter() - t_start, 1)
        details_path = args.details_file
        try:
            os.makedirs(os.path.dirname(os.path.abspath(details_path)), exist_ok=True)
            with open(details_path, "w") as f:
                json.dump(full, f, indent=1)
        except OSError as e:                               # a read-only tree must not cost the line
            details_path = f"(not written: {e.__class__.__name__})"
        report = json.dumps(compact_line(full, details_path))
    finish(report if rank == 0 else None, use_dist)


if __name__ == "__main__":
    main()
