"""Sequence-sharded evaluation runner: the counterpart of ``Evaluator.evaluating`` (networks/engine/eval_manager_mm.py:160-394) for
BASELINE.json configs[4] ("DAVIS-17 + YTB-19 full eval, sequences sharded across 8 x MI355X").

The reference evaluates on one GPU: ``for seq in dataset`` (eval_manager_mm.py:172) with all per-sequence state dropped between
sequences (:376-382).  Sequences are therefore independent units: here they are partitioned over the ranks by longest-processing-time
first on ``frames x objects`` (sharding.lpt_partition), every rank runs its share with its own library handle and no communication,
and ONE all-reduce(SUM) of a small float64 vector (RCCL over xGMI with the "nccl" backend) produces the report.  Within a sequence the
loop is the reference's: frame 0 seeds the pool with its ground truth, every later frame goes through the matching path, the decoder
(here: DynamicPreHead + a fixed linear read-out standing in for the conv decoder, which is out of scope), soft-max, and the memory
policy (eval_loop.MemoryPolicy: never-seen labels, arg-max, entropy -> label 125, pool append every MEM_EVERY frames).

No datasets exist in the build environment: ``make_sequence_set`` generates seeded synthetic clips whose counts mirror DAVIS-17 val
(30 sequences, 480p -> 121x213 maps, <= 3 objects) and YouTube-VOS-19 valid (507 sequences at 6 fps, 145x261 maps, <= 5 objects,
multi-level proxies K in {8, 16, 32}).  The region similarity J (utils/metric.py:3-34 formula) of the predicted masks against the
synthetic ground truth, and the boundary measure F, are accumulated on the device (aoc_mask_jf_accumulate: no mask is ever read back).
"""
import time
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import ops
from . import sharding
from . import synthetic as syn


@dataclass(frozen=True)
class SequenceSpec:
    name: str
    h: int
    w: int
    n_obj: int                  # objects INCLUDING background
    frames: int
    seed: int
    levels: tuple = (16,)       # cluster_num levels of the adaptive-proxy branch
    mem_every: int = 5

    @property
    def cost(self):
        """frames x objects: the quantity the partition balances."""
        return self.frames * self.n_obj

    def clip_config(self):
        return syn.ClipConfig(self.name, self.h, self.w, self.n_obj, max(self.levels), 100, self.frames, self.mem_every)


def make_sequence_set(kind="cfg5", scale=1.0, seed=0) -> List[SequenceSpec]:
    """Synthetic stand-in for the evaluation sets of BASELINE.json configs[4].  ``scale`` < 1 keeps that fraction of each set
    (at least one sequence of each), so that short runs and CPU tests walk the same code path."""
    rng = np.random.RandomState(seed)
    out = []
    if kind in ("cfg5", "davis17"):
        n = max(1, int(round(30 * scale)))
        for i in range(n):          # DAVIS-17 val: 30 sequences, 50..104 frames, 1..3 objects (+ background)
            out.append(SequenceSpec(f"davis{i:02d}", 121, 213, 2 + int(rng.randint(0, 3)), int(rng.randint(50, 105)), 1000 + i, (16,)))
    if kind in ("cfg5", "ytb19"):
        n = max(1, int(round(507 * scale)))
        for i in range(n):          # YouTube-VOS-19 valid: 507 sequences, 6 fps -> 20..36 frames, 1..5 objects (+ background)
            out.append(SequenceSpec(f"ytb{i:03d}", 145, 261, 2 + int(rng.randint(0, 5)), int(rng.randint(20, 37)), 5000 + i, (8, 16, 32)))
    if not out:
        raise ValueError(f"unknown sequence set {kind!r}")
    return out


_STREAMS = {}
SIDE_STREAM_PRIORITY = -1        # the k-means chains' streams (a frame waits for its chain; -1 = high)


def _cached_stream(device, kind, lane, priority=0):
    """The lanes' HIP streams are created once per process and device: torch's caching allocator keeps freed blocks per stream, so a caller that
    runs eval_sharded repeatedly on fresh streams would never get a cached block back (measured: + 10.7 GB reserved per call)."""
    dev = torch.device(device)
    index = dev.index if dev.index is not None else torch.cuda.current_device()      # an index-less "cuda" means the CURRENT device, not device 0
    key = (index, kind, int(lane))
    if key not in _STREAMS:
        _STREAMS[key] = torch.cuda.Stream(device, priority=priority)
    return _STREAMS[key]


class HotPathBackend:
    """One frame on the MI355X: matching (libaoc_hip.so) -> DynamicPreHead -> stand-in read-out -> soft-max.  The conv decoder is out of
    scope; its stand-in ranks the objects by their matching evidence (logit = -12 x the mean of the dense-matching and the widest
    local-window proto-mask channels, i.e. nearest-neighbour label propagation) plus a small seeded linear read-out of the pre-head output,
    which is computed because the real decoder consumes it.  Seeded, so every rank decodes identically."""

    def __init__(self, device, dense_precision=None, ahead=True, lane=0):
        from . import hotpath
        self.hot = hotpath
        self.device = device
        self.dense_precision = dense_precision
        self._heads = {}
        self._runners = {}
        self.max_runners = 4                   # frame-call workspaces kept per lane (LRU)
        self.ahead = bool(ahead)               # False: the per-frame reference-API path (label prep + count read-back + chain on the frame's stream)
        import os
        self.chain_plan = [int(x) for x in os.environ.get("AOC_EVAL_CHAIN_PLAN", "1").split(",") if x.strip()] or [1]   # developer switch
        self.side = None
        self.lane = int(lane)                  # which of the rank's lanes this backend serves (its side stream is kept per lane)
        self._worker, self._pending = None, None

    def _head(self, n_ch):
        if n_ch not in self._heads:
            g = torch.Generator().manual_seed(1234 + n_ch)
            pre = self.hot.DynamicPreHead(in_dim=n_ch, embed_dim=64)
            with torch.no_grad():
                pre.conv.weight.copy_(torch.randn(pre.conv.weight.shape, generator=g) * (2.0 / n_ch) ** 0.5)
                pre.conv.bias.zero_()
            w = (torch.randn(64, generator=g) * 0.02).to(self.device)
            self._heads[n_ch] = (pre.to(self.device), w)
        return self._heads[n_ch]

    def start(self, spec):
        from .eval_loop import MemoryPolicy
        self.spec = spec
        self.mc = self.hot.MatchingConfig(CLUSTER_LEVELS=list(spec.levels) if len(spec.levels) > 1 else None, CLUSTER_NUM=spec.levels[0],
                                          MEM_EVERY=spec.mem_every)
        self.policy = MemoryPolicy(mem_every=spec.mem_every, unc_ratio=1.0)
        self.bias = torch.zeros(spec.n_obj, device=self.device)
        self.rng = np.random.RandomState(spec.seed)
        self.dense_state = {}                   # per sequence: split records of the (append-only) pool, pooled reference heads
        # the pool as ONE resident tensor that grows in place (aocnet.py:128-156 re-stacks the lists every frame), and the adaptive proxies
        # of the frames that will see the same pool, enqueued ahead on a side stream
        cap = spec.frames // max(spec.mem_every, 1) + 2 if spec.mem_every > 0 else 2
        self._pool_emb = torch.empty(cap, spec.h, spec.w, 100, dtype=torch.float32, device=self.device)
        self._pool_lab = torch.empty(cap, spec.h, spec.w, spec.n_obj, dtype=torch.float32, device=self.device)
        self._pool_R = 0
        self._drop_pending()
        self._ahead, self._ahead_rng = [], []
        # ONE C call per frame (aoc_frame_enqueue) where the configuration allows: its workspace is kept per map size / object count
        self.runner = None
        if self.ahead and self.dense_precision in (None, "split") and self.hot.FrameRunner.supported(self.mc, 100, spec.n_obj):
            wcap = (cap + 3) // 4 * 4                      # few distinct workspace sizes per lane
            key = (spec.h, spec.w, spec.n_obj, tuple(spec.levels), wcap)
            if key not in self._runners:
                # a workspace is hundreds of MB (pool records + dense workspace): keep the few most recently used configurations, not one
                # per (map size, object count, capacity) ever seen
                while len(self._runners) >= self.max_runners:
                    self._runners.pop(next(iter(self._runners)))
                self._runners[key] = self.hot.FrameRunner(self.mc, spec.h, spec.w, 100, spec.n_obj, wcap, self.device)
            self.runner = self._runners.pop(key)
            self._runners[key] = self.runner                 # most recently used last
            self.runner.reset()
        if self.ahead and self.side is None:
            self.side = _cached_stream(self.device, "side", self.lane, priority=SIDE_STREAM_PRIORITY)

    def first_frame(self, emb, gt_label):
        self.policy.start(emb, gt_label)

    def _reference_pool(self):
        """policy.reference_pool() without re-stacking: the frames that joined the pool since the last call are appended in place."""
        spec, pol = self.spec, self.policy
        R = len(pol.ref_embeddings)
        if R > self._pool_emb.shape[0]:                     # more joins than planned (ground truth on later frames): grow
            grow = lambda t: torch.cat([t, torch.empty_like(t)], dim=0)
            self._pool_emb, self._pool_lab = grow(self._pool_emb), grow(self._pool_lab)
        for r in range(self._pool_R, R):
            self._pool_emb[r].copy_(pol.ref_embeddings[r])
            self._pool_lab[r].copy_(ops.label_onehot_nearest(pol.ref_mask_confident[r], spec.h, spec.w, spec.n_obj))
        changed = R != self._pool_R
        self._pool_R = R
        return self._pool_emb[:R], self._pool_lab[:R], pol.prev_embedding, ops.label_onehot_nearest(pol.prev_mask, spec.h, spec.w, spec.n_obj), changed

    def _launch_ahead(self, ref_emb, ref_lab):
        """The pool has changed: ONE read-back of the O + 1 row counts for all frames that will see this pool (the reference reads them
        every frame, AEM:263-276), their initial rows drawn from the RandomState in the reference's order (frame, level, object), and their
        k-means chains enqueued on the side stream -- the first frame's alone, the others as one batched chain."""
        spec, mc = self.spec, self.mc
        t = self.policy.frame_idx
        m = spec.mem_every
        n = (-(-t // m) * m - t + 1) if m > 0 else 5
        n = max(1, min(n, spec.frames - t))
        O = spec.n_obj
        self._join_pending(launch=False)                     # a worker may still be drawing for the pool state that just ended
        prep = ops.label_prep(ref_lab.reshape(-1, O))
        counts = prep.counts.cpu().numpy()
        if self._ahead_rng:
            # chains enqueued for a pool that changed earlier than predicted (a later frame carried ground truth) are dropped: their initial
            # rows go back into the stream, so that the draws stay those of the per-frame path (and of the reference: one kmeans2 per frame)
            self.rng.set_state(self._ahead_rng[0])
        self._ahead, self._ahead_rng = [], []
        if int(counts[O]) == 0:
            return                                           # nothing labelled: the per-frame path handles it (AEM:588-589)
        levels = mc.cluster_levels
        kmax = max(levels)
        # the draws of a batch are made right before its chain is launched (numpy.random.RandomState.permutation(n)[:k] per frame, level and
        # object, reproduced by aoc_kmeans_init_rows_draw): the GPU starts on the first frame's chain while the host draws for the others
        sizes, left = [], n
        for b in self.chain_plan:                            # batch sizes of the chains of one pool state (default: the first frame's alone)
            if left <= 0:
                break
            sizes.append(min(b, left))
            left -= sizes[-1]
        if left > 0:
            sizes.append(left)
        # the first batch is drawn and launched now; the draws of the others (a host computation of milliseconds that releases the GIL) run on a
        # worker thread while this thread enqueues the other lanes' frames, and their chains are launched when this lane comes round again
        ready = torch.cuda.Event()                           # the pool and its label prep are final here: all a chain has to wait for
        ready.record()
        self._launch_batch(ops.kmeans_init_rows_draw(self.rng, counts[:O], levels, sizes[0], kmax), ref_emb, ref_lab, prep, ready)
        if len(sizes) > 1:
            if self._worker is None:
                from concurrent.futures import ThreadPoolExecutor
                self._worker = ThreadPoolExecutor(max_workers=1)
            rng, c = self.rng, counts[:O].copy()
            fut = self._worker.submit(lambda: [ops.kmeans_init_rows_draw(rng, c, levels, b, kmax) for b in sizes[1:]])
            self._pending = (fut, ref_emb, ref_lab, prep, ready)

    def _launch_batch(self, drawn, ref_emb, ref_lab, prep, ready):
        rows, states = drawn
        self._ahead_rng += states
        with torch.cuda.stream(self.side):                   # the rows are uploaded on the chain's own stream: nothing of the frame in front of them
            part = [torch.from_numpy(rows[f]).to(self.device, non_blocking=True) for f in range(rows.shape[0])]
        self._ahead += ([self.hot.launch_cluster_proxies(self.mc, ref_emb, ref_lab, part[0], self.side, wait_event=ready, prep=prep)] if len(part) == 1
                        else self.hot.launch_cluster_proxies_batch(self.mc, ref_emb, ref_lab, part, self.side, wait_event=ready, prep=prep))

    def _join_pending(self, launch=True):
        """The draws a worker thread made for the later batches of the current pool state: their generator states join the hand-back list,
        their chains are launched (unless the pool state is over)."""
        if self._pending is None:
            return
        fut, ref_emb, ref_lab, prep, ready = self._pending
        self._pending = None
        for drawn in fut.result():
            if launch:
                self._launch_batch(drawn, ref_emb, ref_lab, prep, ready)
            else:
                self._ahead_rng += drawn[1]

    def _drop_pending(self):
        if getattr(self, "_pending", None) is not None:
            self._pending[0].result()
        self._pending = None

    @torch.no_grad()
    def frame(self, emb):
        """emb [h, w, C] -> predicted label map [H, W] int32 (H = 4 h: the reference's masks live at image resolution)."""
        spec, h, w = self.spec, self.spec.h, self.spec.w
        ref_emb, ref_lab, prev_emb, prev_lab, changed = self._reference_pool()
        if not changed:
            self._join_pending()
        if self.ahead and (changed or not self._ahead):
            self._launch_ahead(ref_emb, ref_lab)
        ahead = self._ahead.pop(0) if self._ahead else None
        if ahead is not None and self._ahead_rng:
            self._ahead_rng.pop(0)
        if self.runner is not None and ahead is not None and ref_emb.shape[0] <= self.runner.call.cap:
            feat, _ = self.runner(ref_emb, ref_lab, prev_emb, prev_lab, emb, self.bias, ahead, pool_key=ref_emb.shape[0])
        else:
            feat, _, _ = self.hot.proto_mask_features(self.mc, ref_emb, ref_lab, prev_emb, prev_lab, emb, self.bias, dense_precision=self.dense_precision,
                                                      dense_state=self.dense_state, rng=self.rng, cluster_ahead=ahead)
        pre, wv = self._head(feat.shape[1])
        y = pre(feat)                                                          # [O, 64, h, w]
        ch = self.hot.channel_slices(self.mc)
        logit = -12.0 * 0.5 * (feat[:, ch["global_fg"]] + feat[:, ch["local"]]) + torch.einsum("ochw,c->ohw", y, wv)
        logit = torch.nn.functional.interpolate(logit[None], size=(4 * h, 4 * w), mode="bilinear", align_corners=True)[0]
        label, _, _ = self.policy.update(emb, torch.softmax(logit, dim=0))
        return label


def _gt_fullres(lab_hw):
    """stride-4 synthetic label map -> image-resolution ground truth (each feature pixel covers a 4x4 block)."""
    return np.kron(lab_hw, np.ones((4, 4), np.int32)).astype(np.int32)


class HostMetric:
    """J of predicted vs ground-truth label maps on the host (sharding.mask_iou_sums over the foreground objects): for CPU-side runs
    of the runner (tests with a stub backend).  Same interface as ops.MaskJF."""

    def __init__(self):
        self.sj, self.n, self.frames = 0.0, 0.0, 0.0

    def add(self, pred, gt, n_obj):
        s, n = sharding.mask_iou_sums(pred, gt, n_obj)
        s0 = 1.0 if int(((pred == 0) | (gt == 0)).sum()) == 0 else int(((pred == 0) & (gt == 0)).sum()) / int(((pred == 0) | (gt == 0)).sum())
        self.sj += s - s0                      # foreground objects only, like the device metric
        self.n += n - 1
        self.frames += 1

    def totals(self):
        return dict(sum_j=self.sj, sum_f=0.0, objects=self.n, frames=self.frames)


def load_sequence(spec: SequenceSpec, device, max_frames: Optional[int] = None):
    """Synthesise one sequence and make it resident on the device: (embeddings [T, h, w, C], image-resolution ground truth [T, H, W] int32).
    Untimed set-up (the stand-in for the dataset loader + backbone, which are out of scope)."""
    cfg = spec.clip_config()
    frames = spec.frames if max_frames is None else min(spec.frames, max_frames)
    clip = syn.make_clip(cfg, spec.seed, frames=frames)
    emb = torch.from_numpy(clip["emb"]).to(device)
    gt = torch.from_numpy(np.stack([_gt_fullres(l) for l in clip["lab"]])).to(device)
    return emb, gt


def _default_metric(device):
    if device.type == "cuda":
        from . import ops
        return ops.MaskJF(device)
    return HostMetric()


def sequence_steps(spec: SequenceSpec, backend, metric, data):
    """The per-sequence loop of eval_manager_mm.py:196-361 as a generator: one ``yield`` after every enqueued frame, so that a caller can
    interleave the frames of several independent sequences (each on its own HIP stream)."""
    emb, gt = data
    backend.start(spec)
    backend.first_frame(emb[0], gt[0])
    yield 0
    for t in range(1, emb.shape[0]):
        pred = backend.frame(emb[t])
        metric.add(pred, gt[t].to(pred.device), spec.n_obj)
        yield t


def run_sequence(spec: SequenceSpec, backend, device, metric=None, max_frames: Optional[int] = None, data=None):
    """One sequence, start to end.  Returns the metric accumulators of sharding.METRIC_FIELDS (sum_iou = sum over frames and
    foreground objects of J, sum_f of F, iou_count = number of (frame, object) pairs).
    data: the (embeddings, ground truth) pair of load_sequence when the caller made it resident beforehand."""
    emb, gt = data if data is not None else load_sequence(spec, device, max_frames)
    frames = emb.shape[0]
    if metric is None:
        metric = _default_metric(device)
    before = metric.totals()
    steps = sequence_steps(spec, backend, metric, (emb, gt))
    next(steps)                                     # pool seeded with frame 0
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in steps:
        pass
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    after = metric.totals()
    return dict(frames=frames - 1, objects=(frames - 1) * (spec.n_obj - 1), gpu_seconds=dt, sum_iou=after["sum_j"] - before["sum_j"],
                sum_f=after["sum_f"] - before["sum_f"], iou_count=after["objects"] - before["objects"])


def run_interleaved(specs_data, device, lanes, backend_factory=None, dense_precision=None, stagger=1):
    """A rank's share with ``lanes`` sequences in flight: every lane owns a HIP stream, a backend (per-sequence state) and a metric
    accumulator, takes the next sequence off the rank's list when it finishes one, and the lanes' frames are enqueued round-robin from
    this one host thread.  The reference-API path reads the O + 1 row counts back once per frame (scipy's initial rows are drawn on the
    host from them): that read-back only waits for its own lane's stream, so the other lanes' frames keep the GPU busy meanwhile.
    specs_data: list of (spec, (embeddings, ground truth)).  Returns the summed accumulators (sharding.METRIC_FIELDS without gpu_seconds)."""
    todo = list(specs_data)
    n_lanes = max(1, min(lanes, len(todo)))
    streams = [_cached_stream(device, "lane", l) for l in range(n_lanes)]
    backends = []
    for l in range(n_lanes):
        b = backend_factory() if backend_factory else HotPathBackend(device, dense_precision, lane=l)
        if backend_factory and hasattr(b, "lane"):
            b.lane = l                  # a factory does not know its lane: without this every backend would share lane 0's cached side stream
        backends.append(b)
    metrics = [_default_metric(device) for _ in range(n_lanes)]
    running = [None] * n_lanes
    frames = objects = 0
    keep = []                                       # the sequences' tensors stay alive until the last lane has drained
    torch.cuda.synchronize(device)
    n_pass = 0
    while True:
        busy = False
        n_pass += 1
        for l in range(n_lanes):
            if n_pass <= l * stagger and todo:
                # lane l takes its first sequence l * stagger passes after lane 0: sequences with the same MEM_EVERY would otherwise reach their
                # pool changes (row-count read-back, host draws, a k-means chain nothing else of the lane overlaps) on the same pass
                busy = True
                continue
            with torch.cuda.stream(streams[l]):
                if running[l] is None and todo:
                    spec, data = todo.pop(0)
                    keep.append(data)
                    frames += data[0].shape[0] - 1
                    objects += (data[0].shape[0] - 1) * (spec.n_obj - 1)
                    running[l] = sequence_steps(spec, backends[l], metrics[l], data)
                if running[l] is not None:
                    busy = True
                    if next(running[l], None) is None:
                        running[l] = None
        if not busy:
            break
    torch.cuda.synchronize(device)
    tot = [m.totals() for m in metrics]
    return dict(frames=frames, objects=objects, sum_iou=sum(t["sum_j"] for t in tot), sum_f=sum(t["sum_f"] for t in tot),
                iou_count=sum(t["objects"] for t in tot))


def eval_sharded(specs: Sequence[SequenceSpec], rank: int, world: int, device, backend=None, metric=None, max_frames=None, barrier=None, lanes=1, stagger=1):
    """Partition ``specs`` over ``world`` ranks (LPT on frames x objects), make this rank's sequences resident on the device, then (after
    ``barrier()`` when given) run them and all-reduce the accumulators.  Returns the job totals plus the load-balance figures (max / mean
    rank time) and ``loop_seconds_max`` = the slowest rank's time for its share, inputs resident; identical on every rank.
    lanes > 1 (GPU only): that many of the rank's sequences are in flight at a time (run_interleaved, lane l starting l * stagger frames after
    lane 0); a rank's time is then its loop time."""
    parts = sharding.lpt_partition([s.cost for s in specs], world)
    mine = parts[rank]
    interleave = lanes > 1 and device.type == "cuda" and backend is None and metric is None
    if backend is None and not interleave:
        backend = HotPathBackend(device)
    data = {i: load_sequence(specs[i], device, max_frames) for i in mine}
    if barrier is not None:
        barrier()
    local = {k: 0.0 for k in sharding.METRIC_FIELDS}
    t0 = time.perf_counter()
    if interleave:
        order = sorted(mine, key=lambda i: -specs[i].cost)           # longest first: the lanes finish together
        r = run_interleaved([(specs[i], data.pop(i)) for i in order], device, lanes, stagger=stagger)
        for k in r:
            local[k] += r[k]
        local["gpu_seconds"] = time.perf_counter() - t0
    else:
        for i in mine:
            r = run_sequence(specs[i], backend, device, metric, max_frames, data=data.pop(i))
            for k in sharding.METRIC_FIELDS:
                local[k] += r[k]
    loop = time.perf_counter() - t0
    dev = device if device.type == "cuda" else None
    tot = sharding.allreduce_metrics(local, device=dev)
    t_max = sharding.allreduce_max(local["gpu_seconds"], device=dev)
    mean_t = tot["gpu_seconds"] / world
    costs = [sum(specs[i].cost for i in p) for p in parts]
    tot.update(sequences=len(specs), ranks=world, rank_seconds_max=t_max, rank_seconds_mean=mean_t, imbalance=t_max / max(mean_t, 1e-12),
               planned_imbalance=max(costs) / max(sum(costs) / world, 1e-12), mean_j=tot["sum_iou"] / max(tot["iou_count"], 1.0),
               mean_f=tot["sum_f"] / max(tot["iou_count"], 1.0), sequences_local=len(mine), loop_seconds_max=sharding.allreduce_max(loop, device=dev))
    return tot
