"""Per-frame hot path: the counterpart of ``AOCNet.before_seghead_process`` (aocnet.py:114-372)
up to the 24-channel proto-mask tensor, plus the FiLM / conditioning gates that
``CalibrationDecoding.forward`` (decoding_module.py:96-149) applies.  The conv / GroupNorm bodies
between the gates are out of scope (they stay ordinary PyTorch-ROCm modules in a full model).

Everything is channel-last and resident in HBM: the reference pool is one [R, h*w, C] tensor that
is never copied or compacted (kernels gather rows through index lists), and every matching
kernel writes straight into its channel slice of the [O, 24, h, w] buffer (aocnet.py:341-358
permutes and concatenates five tensors instead).
"""
from dataclasses import dataclass, field
from typing import List, Optional

import torch
from torch import nn

from . import ops
from .attention import IA_gate
from .conditioning_layer import conditioning_block
from .matching import DEFAULT_CLUSTER_NUM, KMEANS_ITERS, _bias_vec, cluster_proxies


@dataclass
class MatchingConfig:
    """Hot-path constants, named as in configs/resnet101_aocnet.py."""
    MODEL_SEMANTIC_EMBEDDING_DIM: int = 100                      # :65
    MODEL_HEAD_EMBEDDING_DIM: int = 256                          # :66
    MODEL_PRE_HEAD_EMBEDDING_DIM: int = 64                       # :67
    MODEL_REFINE_CHANNELS: int = 64
    MODEL_MULTI_LOCAL_DISTANCE: List[int] = field(default_factory=lambda: [2, 4, 6, 8, 10, 12])   # :70
    MODEL_LOCAL_DOWNSAMPLE: bool = True                          # :71
    MODEL_EPSILON: float = 1e-5                                  # :75
    MODEL_MATCHING_BACKGROUND: bool = True                       # :76
    MODEL_FLOAT16_MATCHING: bool = False                         # :78
    TEST_GLOBAL_ATROUS_RATE: int = 1                             # :125 (tools/eval_net_mm_rpa.py --global_atrous_rate)
    TEST_LOCAL_ATROUS_RATE: int = 1                              # :126
    MEM_EVERY: int = 5                                           # :17
    CLUSTER_NUM: int = DEFAULT_CLUSTER_NUM                       # AEM:232
    CLUSTER_LEVELS: Optional[List[int]] = None                   # multi-level proxies (BASELINE.json configs[2]: [8, 16, 32]); None = [CLUSTER_NUM]
    BETA_PERCENTAGE: float = 0.3

    @property
    def cluster_levels(self):
        """The ``cluster_num`` values (AEM:232) the adaptive-proxy branch runs with, in channel order."""
        return [int(k) for k in self.CLUSTER_LEVELS] if self.CLUSTER_LEVELS else [int(self.CLUSTER_NUM)]

    @property
    def proto_channels(self):
        """in_dim of DynamicPreHead, aocnet.py:43-46 (24 with one cluster level; every further level adds its 2 channels)."""
        n_local = len(self.MODEL_MULTI_LOCAL_DISTANCE)
        c = 2 * (2 + n_local) - 1 + 2 + 2 * (len(self.cluster_levels) - 1)
        return c + (1 + n_local if self.MODEL_MATCHING_BACKGROUND else 0)


# channel layout of the proto-mask tensor (aocnet.py:355-358)
def channel_slices(cfg):
    n = len(cfg.MODEL_MULTI_LOCAL_DISTANCE)
    c2 = 2 * len(cfg.cluster_levels)                   # cluster channels: (centroid, centroid_avg) per level
    s = dict(global_fg=0, cluster=1, proxy=1 + c2, local=2 + c2, local_proxy=2 + c2 + n, prev_mask=2 + c2 + 2 * n)
    if cfg.MODEL_MATCHING_BACKGROUND:
        s.update(local_bg=3 + c2 + 2 * n, global_bg=3 + c2 + 3 * n)
    return s


class ClusterProxiesAhead:
    """Adaptive proxies of ONE frame computed ahead of time on a side stream (launch_cluster_proxies): the k-means
    chain only depends on the reference pool, its labels and the initial rows -- not on the frame before -- so while the
    pool is unchanged (4 of 5 frames with MEM_EVERY = 5) the chain of frame t+1 can run under frame t's other work."""
    __slots__ = ("prep", "table", "sqn", "prep_event", "done_event", "aux", "R", "cluster_sets")


def proxy_table_rows(cfg, n_obj):
    """Rows of one frame's proxy table: levels x objects x (centroid | centroid_avg) x kmax adaptive proxies, then the n_obj k = 1 proxies."""
    levels = cfg.cluster_levels
    return len(levels) * n_obj * 2 * max(levels) + n_obj


def launch_cluster_proxies_batch(cfg, ref_emb, ref_labels, init_rows_list, side_stream=None, wait_event=None, prep=None):
    """Label prep + sticky K + 20 Lloyd iterations + proxy construction (AEM:252-286) of the pool for len(init_rows_list) frames
    that see the same pool, times len(cfg.cluster_levels) levels, enqueued on ``side_stream`` (None = the current stream) without
    any host synchronisation.  All frames x levels x objects advance as segments of ONE k-means chain (same rows, different initial
    rows and K), so each of the ~160 latency-bound launches does the work of all of them; results are bit-identical to separate calls.

    init_rows_list  per frame: int32 device tensor [levels * O, kmax] of segment-local initial rows (level-major; rows drawn like
                    scipy's minit='points': permutation(n_i)[:K_i])
    wait_event      the side stream first waits for it (e.g. the pool append of the previous frame); None = it waits for everything
                    enqueued so far on the caller's current stream (which produced the pool)
    prep            the ops.LabelPrep of ref_labels when the caller has already computed it on its own stream (the evaluation runner
                    reads the row counts back from it): it is reused instead of being computed a second time on the side stream
    Returns one ClusterProxiesAhead per frame, to pass to proto_mask_features(cluster_ahead=...)."""
    F = len(init_rows_list)
    if F > ops.CHAIN_MAX_FRAMES:              # aoc_chain_desc names at most 8 frames: more (MEM_EVERY > 9) go out as several chains over the same label prep
        first = launch_cluster_proxies_batch(cfg, ref_emb, ref_labels, init_rows_list[:ops.CHAIN_MAX_FRAMES], side_stream, wait_event, prep)
        return first + launch_cluster_proxies_batch(cfg, ref_emb, ref_labels, init_rows_list[ops.CHAIN_MAX_FRAMES:], side_stream, wait_event, first[0].prep)
    R, h, w, C = ref_emb.shape
    O = ref_labels.shape[-1]
    hw = h * w
    levels = cfg.cluster_levels
    L, kmax = len(levels), max(levels)
    n_ad = L * O * 2 * kmax
    dev = ref_emb.device
    main = torch.cuda.current_stream()
    side = main if side_stream is None else side_stream
    outs = []
    with torch.cuda.stream(side):
        if side is not main:
            if wait_event is not None:
                side.wait_event(wait_event)
            else:
                side.wait_stream(main)
            # the inputs were allocated on the caller's stream: the caching allocator must not hand their blocks out again
            # while this stream still reads them
            for t in (ref_emb, ref_labels, *init_rows_list):
                t.record_stream(side)
        # the per-frame proxy tables are allocated BEFORE the chain: the caller's stream writes their k = 1 rows as soon as the label
        # prep is done, so they must not share a block with any temporary of the chain that is still running on this stream
        tables = [torch.empty(n_ad + O, C, dtype=torch.float32, device=dev) for _ in range(F)]
        sqns = [torch.empty(n_ad + O, dtype=torch.float32, device=dev) for _ in range(F)]
        pool = ref_emb.reshape(R * hw, C)
        if prep is None:
            prep = ops.label_prep(ref_labels.reshape(R * hw, O))
        elif side is not main:
            for t in (prep.right_bits, prep.wrong_bits, prep.fg_rows, prep.obj_rows, prep.counts, prep.obj_offsets):
                t.record_stream(side)
        prep_event = torch.cuda.Event()
        prep_event.record(side)
        init = init_rows_list[0] if F == 1 else torch.cat([r.reshape(L * O, kmax) for r in init_rows_list], dim=0)
        # ONE C call (aoc_cluster_chain_enqueue): replicated lists with the sticky K, 20 Lloyd iterations, proxy construction, and every frame's
        # proxies scattered into its own table -- what used to be three C calls, a dozen allocations and 2 F table copies (same results)
        ch = ops.cluster_chain(pool, prep, levels, init.reshape(F * L * O, kmax), tables, sqns, KMEANS_ITERS)
        for f in range(F):
            out = ClusterProxiesAhead()
            out.R = R
            out.cluster_sets = None
            out.prep, out.prep_event = prep, prep_event
            out.table, out.sqn = tables[f], sqns[f]
            sl = slice(f * L * O, (f + 1) * L * O)
            out.aux = dict(prep=prep, centroids=ch["centroids"][sl], proxies=ch["proxies"][sl], proxy_sqnorm=ch["proxy_sqnorm"][sl], seg_k=ch["seg_k"][sl],
                           seg_offsets=ch["seg_offsets"], labels=ch["labels"] if F == 1 else None, chain=ch)
            outs.append(out)
        done = torch.cuda.Event()
        done.record(side)
        for out in outs:
            out.done_event = done
    return outs


def prepare_without_clustering(cfg, ref_emb, ref_labels):
    """cfg.MODEL_FLOAT16_MATCHING: the reference's cluster channels are the constant 1 (scipy rejects float16 data), so a frame needs no k-means
    chain -- only the label prep of the pool the matchings see and a proxy table for its k = 1 rows.  Returns the ClusterProxiesAhead that
    FrameRunner / proto_mask_features(cluster_ahead=...) take."""
    R, h, w, C = ref_emb.shape
    O = ref_labels.shape[-1]
    levels = cfg.cluster_levels
    n_ad = len(levels) * O * 2 * max(levels)
    out = ClusterProxiesAhead()
    out.R, out.cluster_sets, out.aux = R, None, None
    out.prep = ops.label_prep(ref_labels.reshape(-1, O))
    out.prep_event = torch.cuda.Event()
    out.prep_event.record()
    out.table = torch.zeros(n_ad + O, C, dtype=torch.float32, device=ref_emb.device)
    out.sqn = torch.full((n_ad + O,), float("inf"), dtype=torch.float32, device=ref_emb.device)
    out.done_event = out.prep_event
    return out


def launch_cluster_proxies(cfg, ref_emb, ref_labels, init_rows_dev, side_stream=None, wait_event=None, prep=None):
    """launch_cluster_proxies_batch for one frame."""
    return launch_cluster_proxies_batch(cfg, ref_emb, ref_labels, [init_rows_dev], side_stream, wait_event, prep)[0]


class IncrementalProxyBank:
    """NON-PARITY mode: the reference re-clusters the WHOLE reference pool for every frame (AEM:263-279), although the pool
    only grows by one frame every MEM_EVERY frames.  The bank clusters each reference frame ONCE, when it joins the pool (the same exact
    k-means chain and the same proxy construction, on that frame's rows only), and matches every later frame against the union of the
    per-frame code books: the min over an object's proxies then runs over R x K proxies instead of K.  k-means work per sequence drops
    from (frames x pool size) to (pool frames x 1) frame-clusterings.  Results differ from the reference's by construction -- as two runs
    of the reference with different numpy seeds differ from each other (tests/test_gpu_round2.py::test_incremental_proxies_accuracy).
    Single cluster level only."""

    def __init__(self, cfg, n_obj, C, capacity_frames, device):
        if cfg.CLUSTER_LEVELS and len(cfg.CLUSTER_LEVELS) > 1:
            raise NotImplementedError("IncrementalProxyBank: one cluster level")
        self.cfg, self.O, self.C, self.cap, self.dev = cfg, n_obj, C, int(capacity_frames), device
        self.K = cfg.cluster_levels[0]
        n_ad = n_obj * 2 * self.cap * self.K
        # layout [object][centroid | centroid_avg][pool frame][K]: the proxies of one set are contiguous
        self.table = torch.zeros(n_ad + n_obj, C, dtype=torch.float32, device=device)
        self.sqn = torch.full((n_ad + n_obj,), float("inf"), dtype=torch.float32, device=device)
        self.R = 0
        self.done_event = None

    def reset(self):
        self.R = 0
        self.sqn.fill_(float("inf"))

    def append(self, frame_emb, frame_labels, init_rows_dev, side_stream=None, wait_event=None, slot=None):
        """Cluster the frame that has just joined the pool (frame_emb [h, w, C], frame_labels [h, w, O], init rows [O, K] segment-local).
        slot: write pool frame `slot` again instead of appending (a benchmark that revisits pool states pays the clustering each time)."""
        r = self.R if slot is None else int(slot)
        assert r < self.cap and r <= self.R, "IncrementalProxyBank: capacity exhausted / slot beyond the pool"
        a = launch_cluster_proxies(self.cfg, frame_emb[None], frame_labels[None], init_rows_dev, side_stream, wait_event)
        main = torch.cuda.current_stream()
        side = main if side_stream is None else side_stream
        O, K, cap = self.O, self.K, self.cap
        with torch.cuda.stream(side):
            n_ad = O * 2 * cap * K
            self.table[:n_ad].view(O, 2, cap, K, self.C)[:, :, r].copy_(a.table[:O * 2 * K].view(O, 2, K, self.C))
            self.sqn[:n_ad].view(O, 2, cap, K)[:, :, r].copy_(a.sqn[:O * 2 * K].view(O, 2, K))
            self.done_event = torch.cuda.Event()
            self.done_event.record(side)
        if r == self.R:
            self.R += 1

    def handle(self, ref_labels):
        """A ClusterProxiesAhead for proto_mask_features(cluster_ahead=...) over the pool's first R frames (ref_labels [R, h, w, O],
        R <= the number of frames in the bank)."""
        R = ref_labels.shape[0]
        assert 1 <= R <= self.R, "the bank holds fewer pool frames"
        out = ClusterProxiesAhead()
        out.R = R
        out.prep = ops.label_prep(ref_labels.reshape(-1, self.O))
        out.prep_event = torch.cuda.Event()
        out.prep_event.record()
        out.done_event = self.done_event
        out.table, out.sqn = self.table, self.sqn
        out.aux = None
        # set (object, centroid | centroid_avg) = the K proxies of each of the R pool frames: contiguous rows of the table
        out.cluster_sets = ([(o * 2 + f) * self.cap * self.K for o in range(self.O) for f in range(2)], [R * self.K] * (2 * self.O),
                            self.O * 2 * self.cap * self.K)
        return out


class PendingCorrelation:
    """The correlation launch of one frame, deferred (proto_mask_features(defer_correlation=True)) so that the frames of several
    sequences in flight can share ONE batched launch (launch_correlations).  `ready` is recorded on the frame's stream when everything
    the launch reads (query, proxy table incl. the k = 1 rows) is final."""
    __slots__ = ("query", "query_split", "table", "sqn", "set_begin", "set_size", "set_off", "set_bias", "feat", "ready", "stream")


def launch_correlations(pending, stream=None, precision="split"):
    """ONE aoc_proxy_corr_min_batched launch for the deferred correlations of several frames (same configuration: map size, objects,
    levels).  Runs on `stream` (default: the current stream) after every frame's `ready` event; returns the event that marks the
    outputs (the cluster / k = 1 proxy channels of every frame's proto-mask tensor) complete -- consumers wait for it."""
    stream = torch.cuda.current_stream() if stream is None else stream
    with torch.cuda.stream(stream):
        for p in pending:
            if p.stream is not stream:
                stream.wait_event(p.ready)
                for t in (p.query, p.table, p.sqn, p.set_bias, p.feat):
                    t.record_stream(stream)
        p0 = pending[0]
        if precision == "split" and all(p.query_split is not None for p in pending):
            for p in pending:
                if p.stream is not stream:
                    p.query_split.records.record_stream(stream)
                    p.query_split.sqnorm.record_stream(stream)
            ops.proxy_corr_min_records([(p.query, p.query_split, p.table, p.sqn, p.set_bias, p.feat) for p in pending], p0.set_begin, p0.set_size,
                                       p0.set_off, True)
        else:
            ops.proxy_corr_min_batched([(p.query, p.table, p.sqn, p.set_bias, p.feat) for p in pending], p0.set_begin, p0.set_size, p0.set_off, True,
                                       precision)
        done = torch.cuda.Event()
        done.record(stream)
    return done


def proto_mask_features(cfg, ref_emb, ref_labels, prev_emb, prev_labels, cur_emb, dis_bias, init_rows=None,
                        cluster_state=None, side_stream=None, dense_state=None, dense_precision=None, cluster_ahead=None,
                        dense_stream=None, defer_correlation=False, rng=None, match_pool=None):
    """All matching branches of one frame -> (features [O, 24, h, w], attention_head [O, 4C], aux).

    ref_emb     [R, h, w, C]  reference pool (channel-last)          ref_labels [R, h, w, O] float one-hot
    prev_emb    [h, w, C]     previous frame embedding               prev_labels [h, w, O]
    cur_emb     [h, w, C]     current (query) frame embedding        dis_bias   [O] or [O,1,1,1]
    init_rows   optional explicit k-means initial rows per object (per level then per object with several cluster levels);
                else drawn like scipy from np.random (or from `rng`, a numpy RandomState)
    cluster_state  optional dict with a device tensor ``init_rows`` [levels * O, kmax] for the host-sync-free pipeline
    side_stream  optional torch.cuda.Stream: the adaptive-proxy branch (k-means: a long chain of small,
                 latency-bound launches) runs there, concurrently with the MFMA-bound dense matching on the
                 current stream; the two join in front of the correlation launch.  Only with cluster_state.
    dense_state  optional dict owned by the caller and kept across the frames of ONE sequence: caches the fp16 split
                 records of the reference pool (only frames appended since the last call are converted) and the pooled
                 reference heads, which only depend on the pool (recomputed when the number of pool frames changes).  The pool
                 must be append-only while the dict lives (the reference's memory policy, eval_manager_mm.py:329-361): a call may see
                 any PREFIX of it; the caller sets dense_state["frames"] = 0 when the pool restarts (next sequence).
    dense_precision  "split" (default, ops.DENSE_PRECISION) or "fp32": see ops.dense_match; the correlation launch follows it (fp16-split
                 kernel with device-side take-over / exact-fp32 kernel).
    dense_stream  optional stream for the dense matching kernel alone (e.g. one created with a HIP CU mask): the call forks
                 to it for that kernel and joins in front of the background maps, so the light kernels of this frame (and
                 of other sequences) are not queued behind the one kernel that fills every CU it may use.
    cluster_ahead  a ClusterProxiesAhead of this frame's pool (launch_cluster_proxies): the adaptive proxies were
                 enqueued earlier on a side stream; this call only waits for them in front of the correlation launch.
    defer_correlation  do not launch the correlation kernel: aux["pending_correlation"] carries it for hotpath.launch_correlations, which
                 batches the frames of several sequences into one launch (the cluster / k = 1 proxy channels of `feat` are complete
                 only after the event that call returns).
    With several cluster levels (cfg.CLUSTER_LEVELS) the cluster channels are (centroid, centroid_avg) per level, in level order.
    The reference's evaluation switches (round 5): cfg.MODEL_FLOAT16_MATCHING (the `.half()` arithmetic of the dense, k = 1 proxy and local
    matchings; cluster channels = the constant 1 the reference degrades to), cfg.MODEL_LOCAL_DOWNSAMPLE, cfg.TEST_LOCAL_ATROUS_RATE, and
    cfg.TEST_GLOBAL_ATROUS_RATE through ``match_pool`` = (embeddings [R, h', w', C], labels [R, h', w', O]) of the pool on the atrous grid
    (ops.atrous_subsample of every pool frame, AEM:533-579): the dense and cluster matchings (and a cluster_ahead) see THAT pool, the pooled
    heads the full one.
    """
    R, h, w, C = ref_emb.shape
    O = ref_labels.shape[-1]
    hw = h * w
    dev = cur_emb.device
    nl = len(cfg.MODEL_MULTI_LOCAL_DISTANCE)
    ch = channel_slices(cfg)
    n_ch = cfg.proto_channels
    bias = _bias_vec(dis_bias, O, dev)
    feat = torch.empty(O, n_ch, h, w, dtype=torch.float32, device=dev)
    base = feat.view(-1)
    obj_stride = n_ch * hw
    f16 = bool(cfg.MODEL_FLOAT16_MATCHING)
    lrate = max(1, int(cfg.TEST_LOCAL_ATROUS_RATE))
    m_emb, m_lab = match_pool if match_pool is not None else (ref_emb, ref_labels)
    pool = m_emb.reshape(-1, C)                                 # what the dense and cluster matchings see
    labels_flat = m_lab.reshape(-1, O)
    query_flat = cur_emb.reshape(hw, C)
    levels = cfg.cluster_levels
    L, kmax = len(levels), max(levels)
    n_ad = L * O * 2 * kmax
    if cluster_ahead is not None and cluster_ahead.cluster_sets is not None:
        n_ad = cluster_ahead.cluster_sets[2]             # IncrementalProxyBank: its own table layout

    # ---- adaptive proxies (k-means, AEM:252-286) + k = 1 proxies (ATT:155-189) in ONE proxy table
    main = torch.cuda.current_stream()
    if cluster_ahead is None and cluster_state is not None:
        # host-sync-free variant: the chain is enqueued here (on side_stream when given) and joined in front of the correlation launch
        cluster_ahead = launch_cluster_proxies(cfg, m_emb, m_lab, cluster_state["init_rows"], side_stream)
    if cluster_ahead is not None:
        assert cluster_ahead.R == R, "cluster_ahead was launched for another pool size"
        main.wait_event(cluster_ahead.prep_event)
        table, sqn, prep, cp = cluster_ahead.table, cluster_ahead.sqn, cluster_ahead.prep, cluster_ahead.aux
        for t in (table, sqn, prep.right_bits, prep.wrong_bits, prep.fg_rows, prep.obj_rows, prep.counts, prep.obj_offsets):
            t.record_stream(main)
    else:
        table = torch.empty(n_ad + O, C, dtype=torch.float32, device=dev)
        sqn = torch.empty(n_ad + O, dtype=torch.float32, device=dev)
        # (float16 matching: no clustering at all -- scipy rejects float16 data and the reference's cluster channels are the constant 1)
        cp = cluster_proxies(pool, labels_flat, levels if cfg.CLUSTER_LEVELS else levels[0], init_rows, rng) if not f16 else None
        prep = cp["prep"] if cp is not None else ops.label_prep(labels_flat)
        if cp is not None:
            table[:n_ad].copy_(cp["proxies"].reshape(-1, C))
            sqn[:n_ad].copy_(cp["proxy_sqnorm"].reshape(-1))
        else:
            sqn[:n_ad].fill_(float("inf"))       # nothing labelled -> every cluster feature is 1

    cached = dense_state.get("ref_pool") if dense_state is not None else None
    k1_copies = []
    if cached is not None and cached[0] == R:
        # the pooled reference heads are a function of the pool alone (ATT:155-170): unchanged since the last frame
        _, ref_pos, ref_neg, ref_sq = cached
        k1_copies = [(ref_pos, table[n_ad:]), (ref_sq, sqn[n_ad:])]       # into the k = 1 rows of this frame's proxy table (by the local-prep launch)
    else:
        ref_pos, ref_neg = ops.masked_mean_pool(ref_emb.reshape(R, hw, C), ref_labels.reshape(R, hw, O), cfg.MODEL_EPSILON, pixel_major=True,
                                                out_pos=table[n_ad:], out_pos_sqnorm=sqn[n_ad:])
        if dense_state is not None:
            dense_state["ref_pool"] = (R, ref_pos.clone(), ref_neg, sqn[n_ad:].clone())
    prev_pos, prev_neg = ops.masked_mean_pool(prev_emb.reshape(1, hw, C), prev_labels.reshape(1, hw, O), cfg.MODEL_EPSILON, pixel_major=True)

    # ---- dense pixel-level matching, AEM:688-817 -> channel 0
    pool_split = None
    if f16:
        dense_precision = "f16"                              # AEM:801-803 on float16 tensors
    if dense_state is not None and (dense_precision or ops.DENSE_PRECISION) == "split" and ops.split_record_bytes(C):
        pool_split = dense_state.get("pool_split")
        done = dense_state.get("frames", 0)
        mhw = pool.shape[0] // R
        if pool_split is None or pool_split.records.shape[0] < R * mhw:
            cap = max(R, dense_state.get("capacity_frames", R)) * mhw
            pool_split = ops.SplitRows()
            pool_split.records = torch.empty(cap, ops.split_record_bytes(C), dtype=torch.uint8, device=dev)
            pool_split.sqnorm = torch.empty(cap, dtype=torch.float32, device=dev)
            pool_split.overflow = torch.zeros(1, dtype=torch.int32, device=dev)
            pool_split.n = cap
            done = 0
        if done < R:
            ops.split_rows(pool[done * mhw:R * mhw], out=pool_split, row0=done * mhw)
        # records of pool frames [0, max(done, R)) are valid; a caller that restarts the pool (new sequence) resets "frames" to 0
        dense_state["pool_split"], dense_state["frames"] = pool_split, max(done, R)
    # the query's split records, ONCE per frame and tile-major: the dense kernel and the correlation kernel both stream them in their MFMA
    # operand layout (the correlation kernel reads the fp32 rows only on its exact-fp32 take-over)
    query_split = None
    if (dense_precision or ops.DENSE_PRECISION) == "split" and ops.split_record_bytes(C) and C == 100 and prep.n_obj <= 16:
        query_split = ops.split_rows(query_flat, overflow=pool_split.overflow if pool_split is not None else None, tiled=True)
    dense_done = None
    if dense_stream is not None:
        main = torch.cuda.current_stream()
        dense_stream.wait_stream(main)
        with torch.cuda.stream(dense_stream):
            ops.dense_match(query_flat, pool, prep, bias, feat, 1, obj_stride, True, precision=dense_precision, query_split=query_split,
                            pool_split=pool_split)
            dense_done = torch.cuda.Event()
            dense_done.record(dense_stream)
        feat.record_stream(dense_stream)
        if query_split is not None:
            query_split.records.record_stream(dense_stream)
            query_split.sqnorm.record_stream(dense_stream)
    else:
        ops.dense_match(query_flat, pool, prep, bias, feat, 1, obj_stride, True, precision=dense_precision, query_split=query_split,
                        pool_split=pool_split)

    # ---- local matching against the previous frame and against its per-pixel proxy map (aocnet.py:255,325-337)
    radii = list(cfg.MODEL_MULTI_LOCAL_DISTANCE)
    prev_flat_labels = prev_labels.reshape(hw, O)
    if cfg.MODEL_LOCAL_DOWNSAMPLE and not f16 and lrate == 1:
        # one launch for the three bilinear down-samples, the (never materialised) proxy map and the label bits; one for both matchings;
        # one for both up-samples into their channel ranges
        H2, W2 = int(h / 2) + 1, int(w / 2) + 1
        q2, p2, pm2, bits2 = ops.local_prep(cur_emb, prev_emb, prev_flat_labels, prev_pos, H2, W2, copies=k1_copies)
        k1_copies = []
        lf = ops.local_window_match_pair(q2, p2, pm2, bits2, radii, bias, O, True)          # [2, O, nl, H2, W2]
        ops.resize_bilinear_planes_grouped(lf.view(2 * O * nl, H2, W2), h, w, base[ch["local"] * hw:], nl, O, (ch["local_proxy"] - ch["local"]) * hw,
                                           obj_stride, hw, 1)
    else:
        # matching.local_matching / local_matching_proxy step by step (AEM:968-1060): full resolution (MODEL_LOCAL_DOWNSAMPLE off), the float16
        # arithmetic (down-samples of float16 tensors), the atrous window stride
        right_prev, _ = ops.label_bits(prev_flat_labels, want_wrong=False)
        proxy_map = ops.label_mix(prev_flat_labels, prev_pos).view(h, w, C)                # aocnet.py:325
        Hm, Wm, qm = h, w, cur_emb
        if cfg.MODEL_LOCAL_DOWNSAMPLE:
            Hm, Wm = int(h / 2) + 1, int(w / 2) + 1
            qm = ops.resize_bilinear_hwc(cur_emb, Hm, Wm, float16=f16)
            right_prev = ops.resize_nearest_bits(right_prev, h, w, Hm, Wm)
        for key, prev_map in (("local", prev_emb), ("local_proxy", proxy_map)):
            pm = ops.resize_bilinear_hwc(prev_map, Hm, Wm, float16=f16) if cfg.MODEL_LOCAL_DOWNSAMPLE else prev_map
            lf = ops.local_window_match(qm, pm, right_prev, radii, bias, O, True, atrous_rate=lrate, float16=f16)        # [O, nl, Hm, Wm]
            ops.resize_bilinear_planes(lf.view(O * nl, Hm, Wm), h, w, base[ch[key] * hw:], hw, 1, inner_count=nl, out_outer_stride=obj_stride)

    # ---- one correlation launch: cluster (2 sets / object / level) + k = 1 proxy (1 set / object), AEM:316-319 + matching.py:2653
    set_begin, set_size, set_off, set_obj = [], [], [], []
    if cluster_ahead is not None and cluster_ahead.cluster_sets is not None:
        set_begin, set_size = list(cluster_ahead.cluster_sets[0]), list(cluster_ahead.cluster_sets[1])
        for o in range(O):
            for f in range(2):
                set_off.append(o * obj_stride + (ch["cluster"] + f) * hw)
                set_obj.append(o)
    else:
        for l, k in enumerate(levels):
            for o in range(O):
                for f in range(2):
                    set_begin.append(((l * O + o) * 2 + f) * kmax)
                    set_size.append(k)                      # slots >= the sticky K carry norm = +inf and are ignored
                    set_off.append(o * obj_stride + (ch["cluster"] + 2 * l + f) * hw)
                    set_obj.append(o)
    for o in range(O):
        set_begin.append(n_ad + o)
        set_size.append(1)
        set_off.append(o * obj_stride + ch["proxy"] * hw)
        set_obj.append(o)
    # per-set bias table (a function of the bias vector alone: kept with the sequence's state while the vector is unchanged).  The cache is
    # keyed on the CALLER's tensor; when _bias_vec had to build a new tensor (float, one element, other dtype / shape) its address says nothing
    # about its value, so nothing is cached then
    src = dis_bias if (torch.is_tensor(dis_bias) and dis_bias.is_cuda and dis_bias.dtype == torch.float32 and dis_bias.numel() == O) else None
    bkey = (src.data_ptr(), src._version, L, O) if src is not None else None
    cached_bias = dense_state.get("set_bias") if (dense_state is not None and bkey is not None) else None
    if cached_bias is not None and cached_bias[0] == bkey:
        set_bias = cached_bias[1]
    else:
        set_bias = bias.repeat_interleave(2).repeat(L) if L > 1 else bias.repeat_interleave(2)
        set_bias = torch.cat([set_bias, bias])
        if dense_state is not None and bkey is not None:
            dense_state["set_bias"] = (bkey, set_bias)
    if cluster_ahead is not None:
        torch.cuda.current_stream().wait_event(cluster_ahead.done_event)   # join: the proxy table is complete
    for src, dst in k1_copies:                        # (no local-prep launch took them along)
        dst.copy_(src)
    pending = None
    if f16:
        # cluster channels: the constant 1 (every cluster distance is the reference's 5e4 padding); k = 1 proxies in the `.half()` arithmetic
        n_cl = 2 * L
        feat[:, ch["cluster"]:ch["cluster"] + n_cl].fill_(1.0)
        ops.proxy_corr_min(query_flat, table, None, [n_ad + o for o in range(O)], [1] * O, [o * obj_stride + ch["proxy"] * hw for o in range(O)], bias, feat, 1,
                           True, float16=True)
    elif defer_correlation:
        pending = PendingCorrelation()
        pending.query, pending.table, pending.sqn, pending.set_bias, pending.feat = query_flat, table, sqn, set_bias, feat
        pending.query_split = query_split
        pending.set_begin, pending.set_size, pending.set_off = set_begin, set_size, set_off
        pending.stream = torch.cuda.current_stream()
        pending.ready = torch.cuda.Event()
        pending.ready.record(pending.stream)
    else:
        # the fp16-split correlation kernel (fp32-equivalent, device-side take-over to exact fp32), also for one frame per launch: with
        # hundreds of proxies (cfg3: 1158, cfg4: 1161) the exact-fp32 MFMA kernel is ten times slower; "fp32" asks for that kernel
        if query_split is not None:
            ops.proxy_corr_min_records([(query_flat, query_split, table, sqn, set_bias, feat)], set_begin, set_size, set_off, True)
        else:
            ops.proxy_corr_min_batched([(query_flat, table, sqn, set_bias, feat)], set_begin, set_size, set_off, True,
                                       "fp32" if (dense_precision or ops.DENSE_PRECISION) == "fp32" else "split")

    if dense_done is not None:
        torch.cuda.current_stream().wait_event(dense_done)           # join: channel 0 (dense) is complete
    # ---- previous-frame mask channel (aocnet.py:356), background maps (AEM:9-23, aocnet.py:349-353) and the attention head (ATT:188) in one launch
    bg = cfg.MODEL_MATCHING_BACKGROUND
    attention_head = ops.proto_finish(feat, hw, obj_stride, ch["local"], nl, ch["local_bg"] if bg else -1, ch["global_fg"], ch["global_bg"] if bg else -1,
                                      ch["prev_mask"], prev_flat_labels, ref_pos.contiguous(), ref_neg.contiguous(), prev_pos, prev_neg)
    return feat, attention_head, dict(cluster=cp, prev_pos=prev_pos, ref_pos=ref_pos, pending_correlation=pending)


class FrameRunner:
    """The frames of ONE sequence through aoc_frame_enqueue (one C call per frame: the counterpart of the single before_seghead_process call of
    aocnet.py:114): same arguments as proto_mask_features with ``cluster_ahead`` (the adaptive proxies come from launch_cluster_proxies[_batch]
    on a side stream), same results bit for bit.  Owns the sequence's device workspace (split records of the pool, pooled reference heads and
    the dense kernel's plan are kept across frames); ``reset()`` when a new sequence starts in it."""

    def __init__(self, cfg, h, w, C, n_obj, capacity_frames, device):
        if not FrameRunner.supported(cfg, C, n_obj):
            raise NotImplementedError("aoc_frame_enqueue does not cover this configuration: use proto_mask_features")
        self.cfg = cfg
        self.call = ops.FrameCall(h, w, C, n_obj, capacity_frames, cfg.MODEL_MULTI_LOCAL_DISTANCE, cfg.cluster_levels, cfg.MODEL_MATCHING_BACKGROUND,
                                  cfg.MODEL_EPSILON, device, float16_matching=cfg.MODEL_FLOAT16_MATCHING, local_downsample=cfg.MODEL_LOCAL_DOWNSAMPLE,
                                  local_atrous_rate=cfg.TEST_LOCAL_ATROUS_RATE, global_atrous_rate=cfg.TEST_GLOBAL_ATROUS_RATE)

    @staticmethod
    def supported(cfg, C, n_obj):
        return ops.FrameCall.supported(C, n_obj, cfg.MODEL_LOCAL_DOWNSAMPLE, cfg.MODEL_FLOAT16_MATCHING, len(cfg.MODEL_MULTI_LOCAL_DISTANCE), len(cfg.cluster_levels))

    def reset(self):
        self.call.reset()

    def match_pool(self, ref_emb, ref_labels, pool_prefix_frames=None):
        """The pool the dense and cluster matchings see: (ref_emb, ref_labels) themselves, or with cfg.TEST_GLOBAL_ATROUS_RATE > 1 their atrous
        sub-samples (AEM:533-579) -- launch_cluster_proxies[_batch] is given THIS pool, the frame call the full one."""
        return self.call.match_pool(ref_emb, ref_labels, pool_prefix_frames)

    def pool_changed(self, valid_frames):
        """The pool's content beyond its first `valid_frames` frames was replaced (a benchmark that revisits pool states): their split records are
        converted again by the next frame.  An append-only pool never needs this."""
        st = self.call.state
        st.records_frames = min(int(st.records_frames), int(valid_frames))

    def __call__(self, ref_emb, ref_labels, prev_emb, prev_labels, cur_emb, dis_bias, cluster_ahead, pool_key, probes=None, pool_prefix_frames=None):
        """pool_key (required): a non-zero value that changes whenever the pool's content changes (an append-only pool may pass its frame count);
        pool_prefix_frames: see ops.FrameCall.__call__ (None = append-only pool)."""
        a = cluster_ahead
        assert a is not None and a.cluster_sets is None and a.R == ref_emb.shape[0], "FrameRunner takes the ClusterProxiesAhead of this frame's pool"
        # (with cfg.TEST_GLOBAL_ATROUS_RATE > 1 the ClusterProxiesAhead -- label prep and k-means chain -- is the one of match_pool(ref_emb, ref_labels))
        O = ref_labels.shape[-1]
        bias = _bias_vec(dis_bias, O, cur_emb.device)
        main = torch.cuda.current_stream()
        for t in (a.table, a.sqn, a.prep.right_bits, a.prep.wrong_bits, a.prep.fg_rows, a.prep.obj_rows, a.prep.counts, a.prep.obj_offsets):
            t.record_stream(main)
        return self.call(ref_emb, ref_labels, prev_emb, prev_labels, cur_emb, bias, a.prep, a.table, a.sqn, a.prep_event, a.done_event, pool_key, probes, pool_prefix_frames)


class DynamicPreHead(nn.Module):
    """decoding_module.py:228-240 (1x1 conv -> GroupNorm(embed_dim / 4) -> ReLU) on the HIP library, with the reference's
    parameter names (``conv``, ``bn``).  ``forward(x)`` mirrors the reference; ``forward(x, cur_emb)`` also performs the
    concatenation of aocnet.py:362 and returns the [O, C + embed_dim, h, w] tensor the decoder consumes."""

    def __init__(self, in_dim=3, embed_dim=100, kernel_size=1):
        super(DynamicPreHead, self).__init__()
        if kernel_size != 1:
            raise NotImplementedError("aoc_amd.DynamicPreHead: kernel_size 1 only (the reference's configs use the default)")
        self.conv = nn.Conv2d(in_dim, embed_dim, kernel_size=kernel_size, stride=1, padding=int((kernel_size - 1) / 2))
        self.bn = nn.GroupNorm(int(embed_dim / 4), embed_dim)
        nn.init.kaiming_normal_(self.conv.weight, mode='fan_out', nonlinearity='relu')

    def forward(self, x, cur_emb=None):
        ops.inference_only("DynamicPreHead", x, cur_emb, *self.parameters())
        return ops.prehead(x, self.conv.weight.detach(), self.conv.bias.detach(), self.bn.num_groups, self.bn.weight.detach(),
                           self.bn.bias.detach(), self.bn.eps, cur_emb)


class CalibrationGates(nn.Module):
    """The ten IA gates and four conditioning blocks of CalibrationDecoding (decoding_module.py:22-84),
    with the reference's attribute names.  ``shapes(h, w)`` lists the activation each one modulates."""

    def __init__(self, cfg=None):
        super().__init__()
        cfg = cfg or MatchingConfig()
        att = cfg.MODEL_SEMANTIC_EMBEDDING_DIM * 4
        in_dim = cfg.MODEL_SEMANTIC_EMBEDDING_DIM + cfg.MODEL_PRE_HEAD_EMBEDDING_DIM
        e, r = cfg.MODEL_HEAD_EMBEDDING_DIM, cfg.MODEL_REFINE_CHANNELS
        b = cfg.BETA_PERCENTAGE
        self.IA1 = IA_gate(att, in_dim)                                   # :22
        self.CLB2 = conditioning_block(e, att, b)                         # :27
        self.CLB3 = conditioning_block(e, att, b)                         # :34
        self.CLB4 = conditioning_block(e * 2, att, b)                     # :41
        self.CLB5 = conditioning_block(e * 2, att, b)                     # :47
        self.IA9 = IA_gate(att + e * 2, e * 2)                            # :52
        self.M1_Reweight_Layer_1 = IA_gate(att, e * 2)                    # :55
        self.M1_Reweight_Layer_2 = IA_gate(att, e * 2)
        self.M1_Reweight_Layer_3 = IA_gate(att, e)
        self.M2_Reweight_Layer_1 = IA_gate(att, e * 2)
        self.M2_Reweight_Layer_2 = IA_gate(att, e * 2)
        self.M2_Reweight_Layer_3 = IA_gate(att, e)
        self.IA10 = IA_gate(att + e + r, e + r)                           # :79
        self.IA11 = IA_gate(att + int(e / 2), int(e / 2))                 # :84
        self.cfg = cfg

    def plan(self, h, w):
        """(module name, channels, map height, map width, extra head width) in call order
        (decoding_module.py:99-148, 162-210).  layer3 has stride 2, so CLB4..M2 see half-size maps."""
        e, r = self.cfg.MODEL_HEAD_EMBEDDING_DIM, self.cfg.MODEL_REFINE_CHANNELS
        in_dim = self.cfg.MODEL_SEMANTIC_EMBEDDING_DIM + self.cfg.MODEL_PRE_HEAD_EMBEDDING_DIM
        h2, w2 = (h + 1) // 2, (w + 1) // 2
        return [("IA1", in_dim, h, w, 0), ("CLB2", e, h, w, 0), ("CLB3", e, h, w, 0), ("CLB4", 2 * e, h2, w2, 0),
                ("CLB5", 2 * e, h2, w2, 0), ("IA9", 2 * e, h2, w2, 2 * e),
                ("M1_Reweight_Layer_1", 2 * e, h2, w2, 0), ("M1_Reweight_Layer_2", 2 * e, h2, w2, 0), ("M1_Reweight_Layer_3", e, h2, w2, 0),
                ("M2_Reweight_Layer_1", 2 * e, h2, w2, 0), ("M2_Reweight_Layer_2", 2 * e, h2, w2, 0), ("M2_Reweight_Layer_3", e, h2, w2, 0),
                ("IA10", e + r, h, w, e + r), ("IA11", int(e / 2), h, w, int(e / 2))]

    @torch.no_grad()
    def forward_batched(self, activations, attention_head, slot=None, probes=None):
        """forward() as ONE C call (aoc_gates_enqueue): the same launches in the same order, bit-identical outputs.  The outputs are PERSISTENT
        buffers owned by this module (one set per `slot`: what a decoder that hands every gate's output straight to the next convolution needs) --
        the next call with the same slot OVERWRITES them, so a caller that keeps a result across calls clones it.  The descriptor list is rebuilt
        when the activation buffers or the weights' storage change."""
        # the descriptors hold raw pointers: of the activations AND of the module weights -- both are part of the key (module.to(), .float(),
        # load_state_dict(assign=True) or a re-assigned parameter give the weights new storage; an in-place update keeps it and needs no rebuild)
        if not hasattr(self, "_param_slots"):
            # the module TREE is fixed after __init__ (walked once, not per frame); the Parameter objects in its slots are not: a re-assigned
            # parameter or load_state_dict(assign=True) puts a new object -- new storage -- into the slot, so the slots are read every call
            self._param_slots = [(m._parameters, n) for m in self.modules() for n in m._parameters]
        key = (tuple((x.data_ptr(), tuple(x.shape)) for x in activations) + (attention_head.shape[1],) +
               tuple(d[n].data_ptr() for d, n in self._param_slots if d[n] is not None))
        if not hasattr(self, "_batches"):
            self._batches = {}
        cached = self._batches.get(slot)                 # one set of output buffers per caller slot (e.g. per sequence in flight)
        if cached is None or cached[0] != key:
            entries = []
            for (name, c, hh, ww, extra), x in zip(self.plan(0, 0), activations):
                mod = getattr(self, name)
                if isinstance(mod, conditioning_block):
                    k_rank = int(mod.CL_1.beta_percentage * x.size()[-1] * x.size()[-2])
                    if k_rank < 1:
                        raise IndexError("conditioning_layer: beta_rank == 0 (the reference fails at beta_val[..., -1])")
                    entries.append((2, x, (mod.mlp_layer.weight, mod.mlp_layer.bias, mod.CL_1.phi_layer.weight.reshape(-1), mod.CL_1.phi_layer.bias,
                                           mod.CL_1.mlp_layer.weight, mod.CL_1.mlp_layer.bias, mod.CL_2.mlp_layer.weight, mod.CL_2.mlp_layer.bias,
                                           mod.CL_3.mlp_layer.weight, mod.CL_3.mlp_layer.bias, k_rank)))
                else:
                    entries.append((1 if extra else 0, x, (mod.IA.weight, mod.IA.bias)))
            cached = (key, ops.GateBatch(entries, activations[0].shape[0], attention_head.shape[1]))
            self._batches[slot] = cached
        return cached[1](attention_head, probes)

    @torch.no_grad()
    def forward(self, activations, attention_head):
        """Applies every gate to its activation (list ordered as ``plan``); returns the modulated list.
        Gates whose head is extended with the inter-object code (IA9/IA10/IA11, decoding_module.py:126-130)
        compute ``px1_delta`` from their own input."""
        out = []
        for (name, c, hh, ww, extra), x in zip(self.plan(0, 0), activations):
            mod = getattr(self, name)
            if isinstance(mod, conditioning_block):
                out.append(mod(x, attention_head))
            else:
                head = attention_head
                if extra:
                    head = ops.head_delta(attention_head, ops.plane_mean(x))        # cat([head, px1.sum(0) - px1]) in one launch
                out.append(mod(x, head))
        return out
