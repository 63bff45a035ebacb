"""Tensor-level front-end of the C ABI: allocates outputs / workspaces as torch device tensors
and passes raw pointers + the current HIP stream to libaoc_hip.so.  PyTorch is plumbing here
(device memory and streams); every computation happens in the HIP library.

All functions require CUDA(HIP) tensors and raise otherwise -- there is no CPU path.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib

PAD_DISTANCE = 5e4
MAX_OBJECTS = 30


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.AocHipError("aoc_amd operators run only on an MI355X device tensor (no CPU fallback); "
                                   f"got a tensor on {t.device}")


def inference_only(what, *tensors):
    """The HIP kernels have no backward: every mirror returns tensors WITHOUT an autograd graph.  Dropping the reference's
    training-time names (IA_gate, conditioning_block, GCT, global_matching, ...) into a training run would therefore train
    nothing, silently.  Raise instead when autograd is recording and an input or parameter wants a gradient."""
    if torch.is_grad_enabled():
        for t in tensors:
            if torch.is_tensor(t) and t.requires_grad:
                raise _lib.AocHipError(f"aoc_amd.{what} is inference-only (no autograd graph is built): call it under torch.no_grad() "
                                       "or detach its inputs; training must use the reference's PyTorch modules")


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    # raw hipStream_t of torch's current stream (the private getters skip ~8 us of Stream-object construction per call;
    # the public API is the fallback when a torch build does not have them)
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return ctypes.c_void_p(_RAW_STREAM(_RAW_DEVICE()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t):
    """float32 + contiguous (the reference hands over permuted views, aocnet.py:149,153,188)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------ labels
def label_bits(labels_flat, want_wrong=True):
    """labels [n, O] float -> (right_bits, wrong_bits) uint32-as-int32 tensors [n]."""
    labels_flat = _f32c(labels_flat)
    _need_gpu(labels_flat)
    n, n_obj = labels_flat.shape
    right = torch.empty(n, dtype=torch.int32, device=labels_flat.device)
    wrong = torch.empty(n, dtype=torch.int32, device=labels_flat.device) if want_wrong else None
    _lib.check(_lib.lib().aoc_label_bits(_p(labels_flat), n, n_obj, _p(right), _p(wrong), _stream()), "aoc_label_bits")
    return right, wrong


class LabelPrep:
    """Result of aoc_label_prep (all device tensors)."""
    __slots__ = ("n", "n_obj", "right_bits", "wrong_bits", "fg_rows", "obj_rows", "counts", "obj_offsets")


def label_prep(labels_flat):
    labels_flat = _f32c(labels_flat)
    _need_gpu(labels_flat)
    n, n_obj = labels_flat.shape
    dev = labels_flat.device
    L = _lib.lib()
    r = LabelPrep()
    r.n, r.n_obj = n, n_obj
    r.right_bits = torch.empty(n, dtype=torch.int32, device=dev)
    r.wrong_bits = torch.empty(n, dtype=torch.int32, device=dev)
    r.fg_rows = torch.empty(n, dtype=torch.int32, device=dev)
    r.obj_rows = torch.empty(n * n_obj, dtype=torch.int32, device=dev)
    r.counts = torch.empty(n_obj + 1, dtype=torch.int32, device=dev)
    r.obj_offsets = torch.empty(n_obj + 1, dtype=torch.int32, device=dev)
    ws = _ws(L.aoc_label_prep_workspace_bytes(n, n_obj), dev)
    _lib.check(L.aoc_label_prep(_p(labels_flat), n, n_obj, _p(r.right_bits), _p(r.wrong_bits), _p(r.fg_rows), _p(r.obj_rows),
                                _p(r.counts), _p(r.obj_offsets), _p(ws), ws.numel(), _stream()), "aoc_label_prep")
    return r


def kmeans_plan(counts, n_seg, cluster_num):
    _need_gpu(counts)
    seg_k = torch.empty(n_seg, dtype=torch.int32, device=counts.device)
    _lib.check(_lib.lib().aoc_kmeans_plan(_p(counts), n_seg, int(cluster_num), _p(seg_k), _stream()), "aoc_kmeans_plan")
    return seg_k


# ------------------------------------------------------------------------------------------ k-means
def kmeans_init_rows_draw(rng, counts, levels, n_frames, kmax=None):
    """Initial rows of the k-means calls of `n_frames` frames that see one pool state, drawn from `rng` (a numpy.random.RandomState, advanced
    in place) exactly as the reference's per-frame kmeans2(minit='points') calls would draw them (aoc_kmeans_init_rows_draw; HOST function).
    counts: rows per object.  Returns (rows int32 [n_frames, len(levels) * n_obj, kmax], states): states[f] = the generator state in front of
    frame f's draws (for rng.set_state when the frames from f on are not used after all)."""
    import numpy as np
    counts = np.ascontiguousarray(np.asarray(counts, dtype=np.int32).reshape(-1))
    lv = np.ascontiguousarray(np.asarray(list(levels), dtype=np.int32))
    kmax = int(max(levels) if kmax is None else kmax)
    st = rng.get_state()
    assert st[0] == "MT19937"
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = ctypes.c_int32(int(st[2]))
    rows = np.empty((int(n_frames), len(lv) * len(counts), kmax), dtype=np.int32)
    snaps = np.empty((int(n_frames), 625), dtype=np.uint32)
    _lib.check(_lib.lib().aoc_kmeans_init_rows_draw(key.ctypes.data, ctypes.byref(pos), counts.ctypes.data, len(counts), lv.ctypes.data, len(lv),
                                                    int(n_frames), kmax, rows.ctypes.data, snaps.ctypes.data), "aoc_kmeans_init_rows_draw")
    rng.set_state((st[0], key, int(pos.value), st[3], st[4]))
    states = [(st[0], snaps[f, :624].copy(), int(snaps[f, 624]), st[3], st[4]) for f in range(int(n_frames))]
    return rows, states


def kmeans_segmented(pool, rows, seg_offsets, seg_k, init_rows, kmax, iters=20, rows_capacity=None, n_rep=1):
    """Segmented k-means, bit-identical to scipy kmeans2 (see include/aoc_hip.h).
    n_rep > 1: the segment lists are n_rep replicas of n_seg / n_rep base segments (kmeans_replicate[_levels]).
    Returns (centroids [S,kmax,C], labels [rows_capacity], cluster_counts [S,kmax])."""
    pool = _f32c(pool)
    _need_gpu(pool, rows, seg_offsets, seg_k, init_rows)
    n_seg = seg_k.numel()
    C = pool.shape[1]
    cap = int(rows.numel() if rows_capacity is None else rows_capacity)
    dev = pool.device
    L = _lib.lib()
    centroids = torch.empty(n_seg, kmax, C, dtype=torch.float32, device=dev)
    labels = torch.empty(cap, dtype=torch.int32, device=dev)
    ccounts = torch.empty(n_seg, kmax, dtype=torch.int32, device=dev)
    ws = _ws(L.aoc_kmeans_workspace_bytes(cap, n_seg, kmax, C), dev)
    init_rows = init_rows.to(torch.int32).contiguous()
    _lib.check(L.aoc_kmeans_segmented_rep(_p(pool), pool.shape[0], C, _p(rows), _p(seg_offsets), _p(seg_k), _p(init_rows), n_seg, int(n_rep), kmax,
                                          int(iters), cap, _p(centroids), _p(labels), _p(ccounts), _p(ws), ws.numel(), _stream()),
               "aoc_kmeans_segmented_rep")
    return centroids, labels, ccounts


def kmeans_replicate(rows, seg_offsets, seg_k, n_rep, rows_capacity=None):
    """Segment lists replicated n_rep times (aoc_kmeans_replicate) -> (rows [n_rep*cap], seg_offsets [n_rep*S+1], seg_k [n_rep*S])."""
    _need_gpu(rows, seg_offsets, seg_k)
    n_seg = seg_k.numel()
    cap = int(rows.numel() if rows_capacity is None else rows_capacity)
    dev = rows.device
    rows_out = torch.empty(n_rep * cap, dtype=torch.int32, device=dev)
    off_out = torch.empty(n_rep * n_seg + 1, dtype=torch.int32, device=dev)
    k_out = torch.empty(n_rep * n_seg, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().aoc_kmeans_replicate(_p(rows), _p(seg_offsets), _p(seg_k), n_seg, int(n_rep), cap, _p(rows_out), _p(off_out), _p(k_out),
                                               _stream()), "aoc_kmeans_replicate")
    return rows_out, off_out, k_out


def kmeans_replicate_levels(rows, seg_offsets, n_seg, n_rep, levels, rows_capacity=None):
    """aoc_kmeans_replicate_levels: n_rep replicas of the segment lists, replica f clustering at K = levels[f % len(levels)] with the
    sticky rule of AEM:268 applied on the device -> (rows [n_rep*cap], seg_offsets [n_rep*S+1], seg_k [n_rep*S])."""
    _need_gpu(rows, seg_offsets)
    cap = int(rows.numel() if rows_capacity is None else rows_capacity)
    dev = rows.device
    lv = np.ascontiguousarray(np.asarray(levels, dtype=np.int32))
    rows_out = rows if n_rep == 1 else torch.empty(n_rep * cap, dtype=torch.int32, device=dev)
    off_out = torch.empty(n_rep * n_seg + 1, dtype=torch.int32, device=dev)
    k_out = torch.empty(n_rep * n_seg, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().aoc_kmeans_replicate_levels(_p(rows), _p(seg_offsets), int(n_seg), int(n_rep), lv.ctypes.data_as(ctypes.c_void_p), int(lv.size),
                                                      cap, _p(rows_out), _p(off_out), _p(k_out), _stream()), "aoc_kmeans_replicate_levels")
    return rows_out, off_out, k_out


def build_proxies(pool, fg_rows, seg_offsets, seg_k, labels, centroids):
    """AEM:280-282 -> (proxies [S,2,kmax,C], proxy_sqnorm [S,2,kmax]); +inf norm = absent proxy."""
    pool = _f32c(pool)
    _need_gpu(pool, fg_rows, seg_offsets, seg_k, labels, centroids)
    n_seg, kmax, C = centroids.shape
    L = _lib.lib()
    proxies = torch.empty(n_seg, 2, kmax, C, dtype=torch.float32, device=pool.device)
    sqnorm = torch.empty(n_seg, 2, kmax, dtype=torch.float32, device=pool.device)
    cap = int(labels.numel())
    ws = _ws(L.aoc_build_proxies_workspace_bytes(cap, n_seg, kmax), pool.device)
    _lib.check(L.aoc_build_proxies(_p(pool), pool.shape[0], C, _p(fg_rows), _p(seg_offsets), _p(seg_k), _p(labels), _p(centroids), n_seg, kmax,
                                   cap, _p(proxies), _p(sqnorm), _p(ws), ws.numel(), _stream()), "aoc_build_proxies")
    return proxies, sqnorm


# ------------------------------------------------------------------------------------------ correlation
def proxy_corr_min(query_flat, proxies, proxy_sqnorm, set_begin, set_size, set_out_offset, set_bias, out, out_pixel_stride,
                   transform=True, float16=False):
    """For every pixel i and set s writes f(min over proxies[set_begin[s] : +set_size[s]] of d(q_i, p)) at
    out.data_ptr()[i*out_pixel_stride + set_out_offset[s]]  (see include/aoc_hip.h).  float16: the reference's `.half()` mode
    (aoc_proxy_corr_min_f16)."""
    query_flat = _f32c(query_flat)
    proxies = _f32c(proxies)
    proxy_sqnorm = _f32c(proxy_sqnorm) if proxy_sqnorm is not None else None
    _need_gpu(query_flat, proxies, proxy_sqnorm, out, set_bias)
    m, C = query_flat.shape
    sb = np.ascontiguousarray(np.asarray(set_begin, dtype=np.int32))
    ss = np.ascontiguousarray(np.asarray(set_size, dtype=np.int32))
    so = np.ascontiguousarray(np.asarray(set_out_offset, dtype=np.int64))
    n_set = sb.size
    assert ss.size == n_set and so.size == n_set
    if set_bias is not None:
        set_bias = _f32c(set_bias)
        assert set_bias.numel() == n_set
    vp = ctypes.c_void_p
    fn = _lib.lib().aoc_proxy_corr_min_f16 if float16 else _lib.lib().aoc_proxy_corr_min
    _lib.check(fn(_p(query_flat), m, C, _p(proxies), _p(proxy_sqnorm), proxies.shape[0], n_set,
                  sb.ctypes.data_as(vp), ss.ctypes.data_as(vp), so.ctypes.data_as(vp), _p(set_bias), _p(out),
                  int(out_pixel_stride), int(bool(transform)), _stream()), "aoc_proxy_corr_min_f16" if float16 else "aoc_proxy_corr_min")
    return out


class _CorrFrame(ctypes.Structure):
    """aoc_corr_frame of include/aoc_hip.h."""
    _fields_ = [("query", ctypes.c_void_p), ("proxies", ctypes.c_void_p), ("proxy_sqnorm", ctypes.c_void_p), ("set_bias", ctypes.c_void_p),
                ("out", ctypes.c_void_p)]


class _CorrFrameRec(ctypes.Structure):
    """aoc_corr_frame_rec of include/aoc_hip.h."""
    _fields_ = [("query", ctypes.c_void_p), ("query_rec", ctypes.c_void_p), ("query_sqnorm", ctypes.c_void_p), ("proxies", ctypes.c_void_p),
                ("proxy_sqnorm", ctypes.c_void_p), ("set_bias", ctypes.c_void_p), ("out", ctypes.c_void_p)]


CORR_PRECISION = {"split": 0, "fp32": 1}
_corr_ws = {}


def _corr_workspace(dev):
    # the 256-byte flag workspace is only touched by kernels of one call, which are ordered on the call's stream: one per (device, stream)
    key = (dev.index, _stream().value)
    ws = _corr_ws.get(key)
    if ws is None:
        ws = _corr_ws[key] = torch.zeros(int(_lib.lib().aoc_proxy_corr_min_batched_workspace_bytes()), dtype=torch.uint8, device=dev)
    return ws


def proxy_corr_min_batched(frames, set_begin, set_size, set_out_offset, transform=True, precision="split"):
    """aoc_proxy_corr_min_batched: the correlation of several frames (of one or of several sequences) in ONE launch.
    frames: sequence of (query_flat [m, C], proxies [n_proxy, C], proxy_sqnorm [n_proxy] or None, set_bias [n_set] or None, out) with the same m, C,
    n_proxy; element (pixel i, set s) of a frame is written at out.data_ptr()[set_out_offset[s] + i] (pixel-contiguous planes)."""
    q0, p0 = frames[0][0], frames[0][1]
    m, C = q0.shape
    n_proxy = p0.shape[0]
    sb = np.ascontiguousarray(np.asarray(set_begin, dtype=np.int32))
    ss = np.ascontiguousarray(np.asarray(set_size, dtype=np.int32))
    so = np.ascontiguousarray(np.asarray(set_out_offset, dtype=np.int64))
    n_set = sb.size
    assert ss.size == n_set and so.size == n_set
    arr = (_CorrFrame * len(frames))()
    keep = []
    for i, (q, p, sq, b, out) in enumerate(frames):
        q, p = _f32c(q), _f32c(p)
        sq = _f32c(sq) if sq is not None else None
        b = _f32c(b) if b is not None else None
        _need_gpu(q, p, sq, b, out)
        assert q.shape == (m, C) and p.shape == (n_proxy, C) and (b is None or b.numel() == n_set)
        keep.append((q, p, sq, b))
        arr[i] = _CorrFrame(q.data_ptr(), p.data_ptr(), sq.data_ptr() if sq is not None else None, b.data_ptr() if b is not None else None,
                            out.data_ptr())
    L = _lib.lib()
    ws = _corr_workspace(q0.device)
    vp = ctypes.c_void_p
    _lib.check(L.aoc_proxy_corr_min_batched(ctypes.cast(arr, vp), len(frames), m, C, n_proxy, n_set, sb.ctypes.data_as(vp), ss.ctypes.data_as(vp),
                                            so.ctypes.data_as(vp), int(bool(transform)), CORR_PRECISION[precision], _p(ws), ws.numel(), _stream()),
               "aoc_proxy_corr_min_batched")
    return [f[4] for f in frames]


class CorrTableCache:
    """Workspace + host key of aoc_proxy_corr_min_records_cached: the tile tables of a frame's correlation passes stay in the workspace, so
    every call is ONE launch whatever the number of proxy tiles.  One per stream / sequence (the calls that share it are stream-ordered)."""

    def __init__(self, device):
        self.ws = torch.zeros(int(_lib.lib().aoc_proxy_corr_min_records_cached_workspace_bytes()), dtype=torch.uint8, device=device)
        self.key = ctypes.c_int64(0)


def proxy_corr_min_records(frames, set_begin, set_size, set_out_offset, transform=True, prepare_only=False, cache=None):
    """aoc_proxy_corr_min_records: proxy_corr_min_batched with every frame's query handed over as the tile-major split records the dense
    kernel consumes for the same frame.  frames: sequence of (query_flat [m, C], query_split (SplitRows, tiled), proxies, proxy_sqnorm or
    None, set_bias or None, out).  cache: a CorrTableCache -> aoc_proxy_corr_min_records_cached (all passes in one launch; same results)."""
    q0, p0 = frames[0][0], frames[0][2]
    m, C = q0.shape
    n_proxy = p0.shape[0]
    sb = np.ascontiguousarray(np.asarray(set_begin, dtype=np.int32))
    ss = np.ascontiguousarray(np.asarray(set_size, dtype=np.int32))
    so = np.ascontiguousarray(np.asarray(set_out_offset, dtype=np.int64))
    n_set = sb.size
    assert ss.size == n_set and so.size == n_set
    arr = (_CorrFrameRec * len(frames))()
    keep = []
    for i, (q, qs, p, sq, b, out) in enumerate(frames):
        q, p = _f32c(q), _f32c(p)
        sq = _f32c(sq) if sq is not None else None
        b = _f32c(b) if b is not None else None
        _need_gpu(q, p, sq, b, out, qs.records, qs.sqnorm)
        assert qs.tiled and qs.n == m, "the records kernel reads aoc_split_rows_tiled records of the whole query"
        assert q.shape == (m, C) and p.shape == (n_proxy, C) and (b is None or b.numel() == n_set)
        keep.append((q, p, sq, b))
        arr[i] = _CorrFrameRec(q.data_ptr(), qs.records.data_ptr(), qs.sqnorm.data_ptr(), p.data_ptr(), sq.data_ptr() if sq is not None else None,
                               b.data_ptr() if b is not None else None, out.data_ptr())
    vp = ctypes.c_void_p
    fn = _lib.lib().aoc_proxy_corr_min_records
    dev = q0.device
    n_frames = len(frames)
    outs = [f[5] for f in frames]

    def launch():
        if cache is not None:
            _lib.check(_lib.lib().aoc_proxy_corr_min_records_cached(ctypes.cast(arr, vp), n_frames, m, C, n_proxy, n_set, sb.ctypes.data_as(vp),
                                                                    ss.ctypes.data_as(vp), so.ctypes.data_as(vp), int(bool(transform)), _p(cache.ws),
                                                                    cache.ws.numel(), ctypes.byref(cache.key), _stream()), "aoc_proxy_corr_min_records_cached")
            return outs
        ws = _corr_workspace(dev)
        _lib.check(fn(ctypes.cast(arr, vp), n_frames, m, C, n_proxy, n_set, sb.ctypes.data_as(vp), ss.ctypes.data_as(vp), so.ctypes.data_as(vp),
                      int(bool(transform)), _p(ws), ws.numel(), _stream()), "aoc_proxy_corr_min_records")
        return outs

    launch.keep = (keep, frames)          # the tensors behind the raw pointers
    if prepare_only:
        return launch                     # bench.py: the argument marshalling stays outside the timed bracket
    return launch()


def dense_match_min(query_flat, pool, prep, obj_bias, out, out_pixel_stride, out_obj_stride, transform=True, float16=False):
    query_flat = _f32c(query_flat)
    pool = _f32c(pool)
    _need_gpu(query_flat, pool, out)
    m, C = query_flat.shape
    n_obj = prep.n_obj
    L = _lib.lib()
    ws = _ws(L.aoc_dense_match_workspace_bytes(m, prep.n, n_obj), pool.device)
    if obj_bias is not None:
        obj_bias = _f32c(obj_bias)
    n_fg = prep.counts[n_obj:n_obj + 1]
    fn = L.aoc_dense_match_min_f16 if float16 else L.aoc_dense_match_min
    _lib.check(fn(_p(query_flat), m, C, _p(pool), _p(prep.fg_rows), _p(n_fg), prep.n, _p(prep.wrong_bits),
                  _p(obj_bias), n_obj, _p(out), int(out_pixel_stride), int(out_obj_stride), int(bool(transform)),
                  _p(ws), ws.numel(), _stream()), "aoc_dense_match_min_f16" if float16 else "aoc_dense_match_min")
    return out


class SplitRows:
    """fp16 split records of embedding rows (aoc_split_rows): records [n, 448] uint8, sqnorm [n], overflow flag [1].  tiled: the records
    in tile-major order (aoc_split_rows_tiled; [ceil(n / 32) * 32, 448] uint8) -- the query side of the dense and correlation kernels."""
    __slots__ = ("records", "sqnorm", "overflow", "n", "tiled")

    def __init__(self):
        self.tiled = False


def split_record_bytes(C):
    return int(_lib.lib().aoc_split_record_bytes(int(C)))


def split_rows(x_flat, out=None, row0=0, overflow=None, tiled=False):
    """x [n, C] fp32 -> SplitRows.  With `out` (a SplitRows with capacity) the records are written at row `row0` of it
    (the reference pool grows in place: only the appended frame is converted).  tiled: tile-major records of the whole x (a query)."""
    x_flat = _f32c(x_flat)
    _need_gpu(x_flat)
    n, C = x_flat.shape
    rb = split_record_bytes(C)
    if rb == 0:
        raise _lib.AocHipError(f"split records need C % 4 == 0 and C <= 100 (got {C})")
    dev = x_flat.device
    if tiled:
        assert out is None and row0 == 0
        out = SplitRows()
        out.tiled = True
        out.records = torch.empty((n + 31) // 32 * 32, rb, dtype=torch.uint8, device=dev)
        out.sqnorm = torch.empty(n, dtype=torch.float32, device=dev)
        out.overflow = overflow if overflow is not None else torch.zeros(1, dtype=torch.int32, device=dev)
        out.n = n
        _lib.check(_lib.lib().aoc_split_rows_tiled(_p(x_flat), n, C, _p(out.records), _p(out.sqnorm), _p(out.overflow), _stream()),
                   "aoc_split_rows_tiled")
        return out
    if out is None:
        out = SplitRows()
        out.records = torch.empty(n, rb, dtype=torch.uint8, device=dev)
        out.sqnorm = torch.empty(n, dtype=torch.float32, device=dev)
        out.overflow = overflow if overflow is not None else torch.zeros(1, dtype=torch.int32, device=dev)
        out.n = n
        row0 = 0
    rec = out.records[row0:row0 + n]
    sq = out.sqnorm[row0:row0 + n]
    assert rec.shape[0] == n
    _lib.check(_lib.lib().aoc_split_rows(_p(x_flat), n, C, _p(rec), _p(sq), _p(out.overflow), _stream()), "aoc_split_rows")
    return out


def dense_prune_stats(reset=True):
    """Developer counters of the coarse-then-rescore dense kernel (aoc_dense_prune_stats_ex): dict(tested, rescored, tiles_rescored, tiles,
    stopped = pairs that ended at the kernel's checkpoint)."""
    import ctypes
    v = (ctypes.c_uint64 * 8)()
    _lib.check(_lib.lib().aoc_dense_prune_stats_ex(v, 1 if reset else 0), "aoc_dense_prune_stats_ex")
    return dict(tested=int(v[0]), rescored=int(v[1]), tiles_rescored=int(v[2]), tiles=int(v[3]), stopped=int(v[4]),
                dev_cycles=[int(v[5]), int(v[6]), int(v[7])])      # development build, AOC_DENSE_DEBUG 4096 / 8192: core-clock sums (tools/bench_dense.py)


def dense_match_min_split(query_flat, query_split, pool, pool_split, prep, obj_bias, out, out_pixel_stride, out_obj_stride, transform=True):
    """aoc_dense_match_min_split: fp16-split matrix pipe with the exact-fp32 kernels as device-side take-over."""
    query_flat = _f32c(query_flat)
    pool = _f32c(pool)
    _need_gpu(query_flat, pool, out, query_split.records, pool_split.records)
    m, C = query_flat.shape
    n_obj = prep.n_obj
    n = prep.n
    assert pool.shape[0] >= n and pool_split.records.shape[0] >= n and query_split.records.shape[0] >= m
    L = _lib.lib()
    ws = _ws(L.aoc_dense_match_split_workspace_bytes(m, n, n_obj), pool.device)
    if obj_bias is not None:
        obj_bias = _f32c(obj_bias)
    # one sticky flag covers both operands
    flag = pool_split.overflow
    if query_split.overflow.data_ptr() != flag.data_ptr():
        flag = torch.maximum(flag, query_split.overflow)
    assert not pool_split.tiled
    _lib.check(L.aoc_dense_match_min_split(_p(query_flat), _p(query_split.records), _p(query_split.sqnorm), int(query_split.tiled), m, C, _p(pool),
                                           _p(pool_split.records),
                                           _p(flag), n, _p(prep.right_bits), _p(prep.wrong_bits), _p(prep.fg_rows), _p(prep.obj_rows),
                                           _p(prep.counts), _p(prep.obj_offsets), _p(obj_bias), n_obj, _p(out), int(out_pixel_stride),
                                           int(out_obj_stride), int(bool(transform)), _p(ws), ws.numel(), _stream()),
               "aoc_dense_match_min_split")
    return out


# "split" = fp16-split matrix pipe with fp32-equivalent products (default); "fp32" = exact-fp32 MFMA everywhere.
DENSE_PRECISION = os.environ.get("AOC_DENSE_PRECISION", "split")


def set_stream_cus(n_cus):
    """aoc_set_stream_cus: how many CUs the streams that launch the matrix kernels may use (a caller that runs them under a HIP CU mask
    says so; 0 = every CU of the device)."""
    _lib.check(_lib.lib().aoc_set_stream_cus(int(n_cus)), "aoc_set_stream_cus")


def dense_match(query_flat, pool, prep, obj_bias, out, out_pixel_stride, out_obj_stride, transform=True, precision=None,
                query_split=None, pool_split=None):
    """Dense matching front door: picks the split-fp16 entry point when the shape supports it (C % 4 == 0, C <= 100,
    <= 16 objects) and the precision mode allows, else the exact-fp32 one.  `query_split` / `pool_split` are optional
    cached SplitRows (otherwise the rows are converted here)."""
    precision = precision or DENSE_PRECISION
    if precision not in ("split", "fp32", "f16"):
        raise ValueError(f"unknown dense precision {precision!r}")
    C = query_flat.shape[1]
    if precision == "f16":            # the reference's use_float16=True arithmetic (AEM:801-803)
        return dense_match_min(query_flat, pool, prep, obj_bias, out, out_pixel_stride, out_obj_stride, transform, float16=True)
    if precision == "fp32" or split_record_bytes(C) == 0 or prep.n_obj > 16:
        return dense_match_min(query_flat, pool, prep, obj_bias, out, out_pixel_stride, out_obj_stride, transform)
    if pool_split is None:
        pool_split = split_rows(pool[:prep.n])
    if query_split is None:
        query_split = split_rows(query_flat, overflow=pool_split.overflow, tiled=True)
    return dense_match_min_split(query_flat, query_split, pool, pool_split, prep, obj_bias, out, out_pixel_stride, out_obj_stride, transform)


# ------------------------------------------------------------------------------------------ local matching + resize
def resize_bilinear_hwc(x, H, W, float16=False):
    x = _f32c(x)
    _need_gpu(x)
    h, w, C = x.shape
    out = torch.empty(H, W, C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().aoc_resize_bilinear_hwc_ex(_p(x), h, w, C, _p(out), H, W, int(bool(float16)), _stream()), "aoc_resize_bilinear_hwc_ex")
    return out


def resize_bilinear_planes(x, H, W, out, out_plane_stride, out_pixel_stride, inner_count=None, out_outer_stride=0):
    """x [P,h,w] -> strided destination (see include/aoc_hip.h); ``out`` may be a view: its data_ptr is the base."""
    x = _f32c(x)
    _need_gpu(x, out)
    P, h, w = x.shape
    inner = P if inner_count is None else int(inner_count)
    _lib.check(_lib.lib().aoc_resize_bilinear_planes(_p(x), P, h, w, _p(out), H, W, inner, int(out_outer_stride), int(out_plane_stride),
                                                     int(out_pixel_stride), _stream()), "aoc_resize_bilinear_planes")
    return out


def atrous_subsample(x, rate):
    """aoc_atrous_subsample: [h, w, X] -> [ceil(h / rate), ceil(w / rate), X] = x[::rate, ::rate] (AEM:533-579, the atrous grid of the reference pool)."""
    x = _f32c(x)
    _need_gpu(x)
    h, w, X = x.shape
    rate = int(rate)
    out = torch.empty((h + rate - 1) // rate, (w + rate - 1) // rate, X, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().aoc_atrous_subsample(_p(x), h, w, X, rate, _p(out), _stream()), "aoc_atrous_subsample")
    return out


def resize_nearest_bits(bits, h, w, H, W):
    _need_gpu(bits)
    out = torch.empty(H * W, dtype=torch.int32, device=bits.device)
    _lib.check(_lib.lib().aoc_resize_nearest_bits(_p(bits), h, w, _p(out), H, W, _stream()), "aoc_resize_nearest_bits")
    return out


def local_window_match(query, prev, right_bits, radii, obj_bias, n_obj, transform=True, atrous_rate=1, float16=False):
    """query, prev [H,W,C] -> [n_obj, len(radii), H, W] (channel order [max, r_0, ...]); atrous_rate / float16 as in the reference call."""
    query, prev = _f32c(query), _f32c(prev)
    _need_gpu(query, prev, right_bits)
    H, W, C = query.shape
    radii = np.ascontiguousarray(np.asarray(radii, dtype=np.int32))
    out = torch.empty(n_obj, radii.size, H, W, dtype=torch.float32, device=query.device)
    if obj_bias is not None:
        obj_bias = _f32c(obj_bias)
    _lib.check(_lib.lib().aoc_local_window_match_ex(_p(query), _p(prev), _p(right_bits), H, W, C, radii.ctypes.data_as(ctypes.c_void_p),
                                                    int(radii.size), _p(obj_bias), n_obj, _p(out), int(bool(transform)), int(atrous_rate),
                                                    int(bool(float16)), _stream()), "aoc_local_window_match_ex")
    return out


def local_prep(cur_emb, prev_emb, prev_labels_flat, prev_pos, H2, W2, obj_bias=None, n_pair_sets=0, set_bias_out=None, copies=()):
    """aoc_local_prep: the half-resolution operands of both local matchings in one launch -> (q2, p2, pm2 [H2, W2, C], bits2 [H2 * W2]).
    copies: up to two (src, dst) float tensor pairs copied by the same launch; set_bias_out [n_pair_sets + O] is filled from obj_bias."""
    cur_emb, prev_emb, prev_labels_flat, prev_pos = _f32c(cur_emb), _f32c(prev_emb), _f32c(prev_labels_flat), _f32c(prev_pos)
    _need_gpu(cur_emb, prev_emb, prev_labels_flat, prev_pos, obj_bias, set_bias_out)
    h, w, C = cur_emb.shape
    n_obj = prev_labels_flat.shape[-1]
    dev = cur_emb.device
    q2 = torch.empty(H2, W2, C, dtype=torch.float32, device=dev)
    p2, pm2 = torch.empty_like(q2), torch.empty_like(q2)
    bits2 = torch.empty(H2 * W2, dtype=torch.int32, device=dev)
    cp = [(None, None, 0), (None, None, 0)]
    for i, (src, dst) in enumerate(copies):
        assert src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype == torch.float32 and src.numel() == dst.numel()
        cp[i] = (src, dst, src.numel())
    if obj_bias is not None:
        obj_bias = _f32c(obj_bias)
    _lib.check(_lib.lib().aoc_local_prep(_p(cur_emb), _p(prev_emb), _p(prev_labels_flat), _p(prev_pos), h, w, C, n_obj, _p(q2), _p(p2), _p(pm2), _p(bits2),
                                         int(H2), int(W2), _p(obj_bias), int(n_pair_sets), _p(set_bias_out), _p(cp[0][0]), _p(cp[0][1]), cp[0][2],
                                         _p(cp[1][0]), _p(cp[1][1]), cp[1][2], _stream()), "aoc_local_prep")
    return q2, p2, pm2, bits2


def local_window_match_pair(query, prev_a, prev_b, right_bits, radii, obj_bias, n_obj, transform=True):
    """aoc_local_window_match_pair -> [2, n_obj, len(radii), H, W]: local matching against prev_a and prev_b in one launch."""
    query, prev_a, prev_b = _f32c(query), _f32c(prev_a), _f32c(prev_b)
    _need_gpu(query, prev_a, prev_b, right_bits)
    H, W, C = query.shape
    radii = np.ascontiguousarray(np.asarray(radii, dtype=np.int32))
    out = torch.empty(2, n_obj, radii.size, H, W, dtype=torch.float32, device=query.device)
    if obj_bias is not None:
        obj_bias = _f32c(obj_bias)
    _lib.check(_lib.lib().aoc_local_window_match_pair(_p(query), _p(prev_a), _p(prev_b), _p(right_bits), H, W, C, radii.ctypes.data_as(ctypes.c_void_p),
                                                      int(radii.size), _p(obj_bias), n_obj, _p(out[0]), _p(out[1]), int(bool(transform)), _stream()),
               "aoc_local_window_match_pair")
    return out


def resize_bilinear_planes_grouped(x, H, W, out, inner_count, outer_count, group_stride, outer_stride, plane_stride, pixel_stride=1):
    """aoc_resize_bilinear_planes_grouped: x [P, h, w], plane p = (group, outer, inner) -> strided destination."""
    x = _f32c(x)
    _need_gpu(x, out)
    P, h, w = x.shape
    _lib.check(_lib.lib().aoc_resize_bilinear_planes_grouped(_p(x), P, h, w, _p(out), H, W, int(inner_count), int(outer_count), int(group_stride),
                                                             int(outer_stride), int(plane_stride), int(pixel_stride), _stream()),
               "aoc_resize_bilinear_planes_grouped")
    return out


def proto_finish(feat, hw, obj_stride, ch_local, n_local, ch_local_bg, ch_global, ch_global_bg, ch_prev_mask, prev_labels_flat,
                 ref_pos=None, ref_neg=None, prev_pos=None, prev_neg=None):
    """aoc_proto_finish: background channels, previous-mask channel and (when the four pooled heads are given) the attention head
    [O, 4C] of one frame in one launch.  Returns the head (or None)."""
    _need_gpu(feat, prev_labels_flat, ref_pos, ref_neg, prev_pos, prev_neg)
    n_obj = feat.shape[0]
    head, C = None, 0
    if ref_pos is not None:
        C = ref_pos.shape[1]
        head = torch.empty(n_obj, 4 * C, dtype=torch.float32, device=feat.device)
        for t in (ref_pos, ref_neg, prev_pos, prev_neg):
            assert t.is_contiguous() and t.shape == (n_obj, C)
    if prev_labels_flat is not None:
        prev_labels_flat = _f32c(prev_labels_flat)
    _lib.check(_lib.lib().aoc_proto_finish(_p(feat), n_obj, int(hw), int(obj_stride), int(ch_local), int(n_local), int(ch_local_bg), int(ch_global),
                                           int(ch_global_bg), int(ch_prev_mask), _p(prev_labels_flat), _p(ref_pos), _p(ref_neg), _p(prev_pos), _p(prev_neg),
                                           int(C), _p(head), _stream()), "aoc_proto_finish")
    return head


class _FrameDesc(ctypes.Structure):
    """aoc_frame_desc of include/aoc_hip.h."""
    _fields_ = [("h", ctypes.c_int32), ("w", ctypes.c_int32), ("C", ctypes.c_int32), ("n_obj", ctypes.c_int32),
                ("R", ctypes.c_int32), ("R_capacity", ctypes.c_int32),
                ("n_radii", ctypes.c_int32), ("radii", ctypes.c_int32 * 8),
                ("n_levels", ctypes.c_int32), ("levels", ctypes.c_int32 * 8),
                ("kmax", ctypes.c_int32), ("matching_background", ctypes.c_int32), ("n_adaptive", ctypes.c_int32), ("epsilon", ctypes.c_float),
                ("pool_prefix_frames", ctypes.c_int32), ("stream_cus", ctypes.c_int32),
                ("float16_matching", ctypes.c_int32), ("local_downsample", ctypes.c_int32), ("local_atrous_rate", ctypes.c_int32), ("match_hw", ctypes.c_int32),
                ("pool_key", ctypes.c_int64),
                ("ref_emb", ctypes.c_void_p), ("ref_labels", ctypes.c_void_p), ("match_emb", ctypes.c_void_p), ("prev_emb", ctypes.c_void_p), ("prev_labels", ctypes.c_void_p),
                ("cur_emb", ctypes.c_void_p), ("dis_bias", ctypes.c_void_p),
                ("right_bits", ctypes.c_void_p), ("wrong_bits", ctypes.c_void_p), ("fg_rows", ctypes.c_void_p), ("obj_rows", ctypes.c_void_p),
                ("counts", ctypes.c_void_p), ("obj_offsets", ctypes.c_void_p),
                ("proxy_table", ctypes.c_void_p), ("proxy_sqnorm", ctypes.c_void_p), ("prep_ready", ctypes.c_void_p), ("proxies_ready", ctypes.c_void_p),
                ("feat", ctypes.c_void_p), ("head", ctypes.c_void_p), ("probe", ctypes.c_void_p * 6)]


class _ChainDesc(ctypes.Structure):
    """aoc_chain_desc of include/aoc_hip.h."""
    _fields_ = [("C", ctypes.c_int32), ("n_obj", ctypes.c_int32), ("n_frames", ctypes.c_int32), ("n_levels", ctypes.c_int32), ("levels", ctypes.c_int32 * 8),
                ("kmax", ctypes.c_int32), ("iters", ctypes.c_int32), ("reserved0", ctypes.c_int32),
                ("pool_rows", ctypes.c_int64), ("rows_capacity", ctypes.c_int64),
                ("pool", ctypes.c_void_p), ("fg_rows", ctypes.c_void_p), ("obj_rows", ctypes.c_void_p), ("obj_offsets", ctypes.c_void_p),
                ("init_rows", ctypes.c_void_p), ("tables", ctypes.c_void_p * 8), ("sqnorms", ctypes.c_void_p * 8)]


CHAIN_MAX_FRAMES = 8      # frames one aoc_chain_desc names (tables[8] / sqnorms[8])


def cluster_chain(pool, prep, levels, init_rows, tables, sqnorms, iters=20):
    """aoc_cluster_chain_enqueue: the k-means chain of len(tables) frames that see one pool state -- replicated lists with sticky K, 20 Lloyd
    iterations, proxy construction, every frame's proxies scattered into ITS table -- as ONE C call out of one workspace (on the current stream,
    no host synchronisation).  pool [rows, C]; prep = LabelPrep of the pool's labels; init_rows int32 [F * L * O, kmax]; tables[f] [L*O*2*kmax + O, C],
    sqnorms[f] [L*O*2*kmax + O].  Returns a dict of views into the workspace (centroids, labels, cluster_counts, proxies, proxy_sqnorm, seg_k,
    seg_offsets), valid until the workspace is reused."""
    pool = _f32c(pool)
    _need_gpu(pool, init_rows, *tables, *sqnorms)
    F, L, O, C = len(tables), len(levels), prep.n_obj, pool.shape[1]
    kmax = int(max(levels))
    init_rows = init_rows.to(torch.int32).contiguous()
    if not (1 <= F <= CHAIN_MAX_FRAMES and 1 <= L <= 8 and len(sqnorms) == F and init_rows.numel() == F * L * O * kmax):
        raise _lib.AocHipError("cluster_chain: 1..8 frames, 1..8 levels, one squared-norm array per table, init_rows [F * L * O, kmax]")
    d = _ChainDesc()
    d.C, d.n_obj, d.n_frames, d.n_levels, d.kmax, d.iters = C, O, F, L, kmax, int(iters)
    for i, k in enumerate(levels):
        d.levels[i] = int(k)
    d.pool_rows, d.rows_capacity = pool.shape[0], prep.obj_rows.numel()
    d.pool, d.fg_rows, d.obj_rows, d.obj_offsets, d.init_rows = pool.data_ptr(), prep.fg_rows.data_ptr(), prep.obj_rows.data_ptr(), prep.obj_offsets.data_ptr(), init_rows.data_ptr()
    for f in range(F):
        assert tables[f].is_contiguous() and tables[f].dtype == torch.float32 and tables[f].shape[0] >= L * O * 2 * kmax
        d.tables[f], d.sqnorms[f] = tables[f].data_ptr(), sqnorms[f].data_ptr()
    lib = _lib.lib()
    ws = _ws(lib.aoc_cluster_chain_workspace_bytes(ctypes.byref(d)), pool.device)
    _lib.check(lib.aoc_cluster_chain_enqueue(ctypes.byref(d), _p(ws), ws.numel(), _stream()), "aoc_cluster_chain_enqueue")
    off = (ctypes.c_int64 * 7)()
    _lib.check(lib.aoc_cluster_chain_layout(ctypes.byref(d), off), "aoc_cluster_chain_layout")
    S, cap = F * L * O, F * L * prep.obj_rows.numel()

    def view(i, n, dtype, shape):
        return ws[off[i]:off[i] + n * 4].view(dtype).view(shape)
    return dict(workspace=ws, keep=(pool, init_rows), centroids=view(0, S * kmax * C, torch.float32, (S, kmax, C)), labels=view(1, cap, torch.int32, (cap,)),
                cluster_counts=view(2, S * kmax, torch.int32, (S, kmax)), proxies=view(3, S * 2 * kmax * C, torch.float32, (S, 2, kmax, C)),
                proxy_sqnorm=view(4, S * 2 * kmax, torch.float32, (S, 2, kmax)), seg_k=view(5, S, torch.int32, (S,)),
                seg_offsets=view(6, S + 1, torch.int32, (S + 1,)))


class _GateDesc(ctypes.Structure):
    """aoc_gate_desc of include/aoc_hip.h."""
    _fields_ = [("kind", ctypes.c_int32), ("channels", ctypes.c_int32), ("k_rank", ctypes.c_int32), ("reserved", ctypes.c_int32), ("hw", ctypes.c_int64),
                ("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("w", ctypes.c_void_p), ("b", ctypes.c_void_p),
                ("phi_w", ctypes.c_void_p), ("phi_b", ctypes.c_void_p), ("w1", ctypes.c_void_p), ("b1", ctypes.c_void_p), ("w2", ctypes.c_void_p),
                ("b2", ctypes.c_void_p), ("w3", ctypes.c_void_p), ("b3", ctypes.c_void_p), ("probe", ctypes.c_void_p * 4)]


class GateBatch:
    """aoc_gates_enqueue: a fixed list of gates (modules + the activations they modulate + persistent outputs) applied by ONE C call per frame.
    entries: (kind, x, params) with params = (w, b) for kinds 0 / 1 and (w, b, phi_w, phi_b, w1, b1, w2, b2, w3, b3, k_rank) for kind 2."""

    def __init__(self, entries, n_obj, head_dim):
        self.n, self.n_obj, self.D = len(entries), int(n_obj), int(head_dim)
        self.arr = (_GateDesc * self.n)()
        self.keep, self.outs = [], []
        for i, (kind, x, prm) in enumerate(entries):
            assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[0] == n_obj
            y = torch.empty_like(x)
            ts = [_f32c(t.detach()) for t in prm[:10]]
            self.keep.append((x, ts))
            self.outs.append(y)
            d = self.arr[i]
            d.kind, d.channels, d.hw = int(kind), int(x.shape[1]), int(x.numel() // (x.shape[0] * x.shape[1]))
            d.x, d.y, d.w, d.b = x.data_ptr(), y.data_ptr(), ts[0].data_ptr(), ts[1].data_ptr()
            if kind == 2:
                d.phi_w, d.phi_b, d.w1, d.b1, d.w2, d.b2, d.w3, d.b3 = [t.data_ptr() for t in ts[2:10]]
                d.k_rank = int(prm[10])
        L = _lib.lib()
        self.ws = torch.empty(max(16, int(L.aoc_gates_workspace_bytes(ctypes.cast(self.arr, ctypes.c_void_p), self.n, self.n_obj, self.D))),
                              dtype=torch.uint8, device=entries[0][1].device)

    def __call__(self, head, probes=None):
        """probes (measurement only): per gate a list of four raw hipEvent_t handles or None (aoc_gate_desc.probe)."""
        head = _f32c(head)
        _need_gpu(head)
        assert tuple(head.shape) == (self.n_obj, self.D)
        for i in range(self.n):
            for k in range(4):
                self.arr[i].probe[k] = probes[i][k] if probes is not None else None
        _lib.check(_lib.lib().aoc_gates_enqueue(ctypes.cast(self.arr, ctypes.c_void_p), self.n, _p(head), self.n_obj, self.D, _p(self.ws), self.ws.numel(),
                                                _stream()), "aoc_gates_enqueue")
        return self.outs


class _SeqState(ctypes.Structure):
    """aoc_seq_state of include/aoc_hip.h."""
    _fields_ = [("initialised", ctypes.c_int64), ("records_frames", ctypes.c_int64), ("ref_pool_key", ctypes.c_int64), ("plan_key", ctypes.c_int64),
                ("plan_rows", ctypes.c_int64), ("corr_tables_key", ctypes.c_int64)]


class FrameCall:
    """aoc_frame_enqueue for the frames of ONE sequence: owns the sequence's device workspace and host state record.
    supported(...) says whether the one-call path covers a configuration (otherwise hotpath.proto_mask_features drives the individual calls)."""

    @staticmethod
    def supported(C, n_obj, local_downsample, float16_matching, n_radii, n_levels):
        """Round 5: every switch of the reference's evaluation CLI / config is covered (float16 matching, local matching with or without the
        down-sample, the atrous rates, up to 30 objects); what remains outside is another embedding width and the developer's exact-fp32 mode."""
        return C == 100 and n_obj <= MAX_OBJECTS and n_radii <= 8 and n_levels <= 8 and DENSE_PRECISION == "split"

    def __init__(self, h, w, C, n_obj, capacity_frames, radii, levels, matching_background, epsilon, device, float16_matching=False, local_downsample=True,
                 local_atrous_rate=1, global_atrous_rate=1):
        L = _lib.lib()
        self.h, self.w, self.C, self.n_obj, self.cap = int(h), int(w), int(C), int(n_obj), int(capacity_frames)
        self.n_ch = int(L.aoc_frame_channels(len(radii), len(levels), int(bool(matching_background))))
        nbytes = int(L.aoc_frame_workspace_bytes(self.h, self.w, self.C, self.n_obj, self.cap, len(radii), len(levels)))
        if nbytes == 0:
            raise _lib.AocHipError("aoc_frame_workspace_bytes: unsupported configuration")
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.state = _SeqState()
        d = self.desc = _FrameDesc()
        d.h, d.w, d.C, d.n_obj, d.R_capacity = self.h, self.w, self.C, self.n_obj, self.cap
        d.n_radii, d.n_levels, d.kmax = len(radii), len(levels), max(levels)
        for i, r in enumerate(radii):
            d.radii[i] = int(r)
        for i, k in enumerate(levels):
            d.levels[i] = int(k)
        d.matching_background = int(bool(matching_background))
        d.n_adaptive = len(levels) * self.n_obj * 2 * max(levels)
        d.epsilon = float(epsilon)
        d.float16_matching, d.local_downsample, d.local_atrous_rate = int(bool(float16_matching)), int(bool(local_downsample)), int(local_atrous_rate)
        # TEST_GLOBAL_ATROUS_RATE > 1: the dense and cluster matchings see every pool frame on the atrous grid (AEM:533-579); the sub-sampled pool is
        # kept here, one aoc_atrous_subsample per frame that joins (append-only), and handed over as match_emb
        self.grate = int(global_atrous_rate)
        self.match_hw = ((self.h + self.grate - 1) // self.grate) * ((self.w + self.grate - 1) // self.grate) if self.grate > 1 else 0
        self.match_emb = torch.empty(self.cap, self.match_hw, self.C, dtype=torch.float32, device=device) if self.grate > 1 else None
        self.match_frames = 0
        self.device = device

    def reset(self):
        """A new sequence starts in this workspace (eval_manager_mm.py:376-382)."""
        ctypes.memset(ctypes.byref(self.state), 0, ctypes.sizeof(self.state))
        self.match_frames = 0

    def match_pool(self, ref_emb, ref_labels, pool_prefix_frames=None):
        """TEST_GLOBAL_ATROUS_RATE > 1: (embeddings [R, h', w', C], labels [R, h', w', O]) of the pool on the atrous grid -- what the label prep and the
        k-means chain of the frame have to be computed from.  The embeddings are kept per sequence (only frames that joined are sub-sampled);
        rate 1: the arguments themselves."""
        if self.grate <= 1:
            return ref_emb, ref_labels
        R = ref_emb.shape[0]
        done = min(self.match_frames, R if pool_prefix_frames is None else int(pool_prefix_frames))
        hh, ww = (self.h + self.grate - 1) // self.grate, (self.w + self.grate - 1) // self.grate
        for r in range(done, R):
            _lib.check(_lib.lib().aoc_atrous_subsample(_p(ref_emb[r]), self.h, self.w, self.C, self.grate, _p(self.match_emb[r]), _stream()), "aoc_atrous_subsample")
        self.match_frames = R
        labs = torch.stack([atrous_subsample(ref_labels[r], self.grate) for r in range(R)])
        return self.match_emb[:R].view(R, hh, ww, self.C), labs

    def __call__(self, ref_emb, ref_labels, prev_emb, prev_labels, cur_emb, dis_bias, prep, table, sqn, prep_event=None, done_event=None, pool_key=None,
                 probes=None, pool_prefix_frames=None, stream_cus=0):
        """ref_emb [R, h, w, C] / ref_labels [R, h, w, O] contiguous fp32 views of the resident pool; prep = LabelPrep of ref_labels; table / sqn = this
        frame's proxy table (adaptive rows written by the k-means chain).  Returns (feat [O, n_ch, h, w], head [O, 4C]).
        pool_key (REQUIRED, != 0): identifies the pool's content -- it keys the pooled reference heads and the dense kernel's plan; an append-only pool
        may pass its frame count.  pool_prefix_frames: leading pool frames unchanged since the previous call (their split records are kept); None =
        the pool is append-only (everything converted so far stays valid); a caller that REPLACES pool frames passes the first replaced index."""
        if pool_key is None or int(pool_key) == 0:
            raise _lib.AocHipError("FrameCall: pool_key is required (a non-zero value that changes whenever the pool's content changes)")
        _need_gpu(ref_emb, ref_labels, prev_emb, prev_labels, cur_emb, dis_bias, table, sqn)
        for t in (ref_emb, ref_labels, prev_emb, prev_labels, cur_emb, dis_bias, table, sqn):
            assert t.dtype == torch.float32 and t.is_contiguous(), "aoc_frame_enqueue takes contiguous float32 tensors"
        R = ref_emb.shape[0]
        d = self.desc
        assert R <= self.cap and tuple(cur_emb.shape) == (self.h, self.w, self.C) and ref_labels.shape[-1] == self.n_obj and dis_bias.numel() == self.n_obj
        assert table.shape[0] == d.n_adaptive + self.n_obj and prep.n == R * (self.match_hw if self.grate > 1 else self.h * self.w), \
            "prep = the label prep of the pool the matchings see (match_pool(...) with a global atrous rate)"
        if self.grate > 1:
            assert self.match_frames >= R, "call match_pool(ref_emb, ref_labels) first: it keeps the sub-sampled pool the matchings read"
            d.match_hw, d.match_emb = self.match_hw, self.match_emb.data_ptr()
        else:
            d.match_hw, d.match_emb = 0, None
        feat = torch.empty(self.n_obj, self.n_ch, self.h, self.w, dtype=torch.float32, device=self.device)
        head = torch.empty(self.n_obj, 4 * self.C, dtype=torch.float32, device=self.device)
        d.R = R
        d.pool_key = int(pool_key)
        d.pool_prefix_frames = int(R if pool_prefix_frames is None else pool_prefix_frames)
        d.stream_cus = int(stream_cus)              # > 0: this call's stream runs under a CU mask of that many CUs (else: aoc_set_stream_cus)
        d.ref_emb, d.ref_labels, d.prev_emb, d.prev_labels, d.cur_emb = ref_emb.data_ptr(), ref_labels.data_ptr(), prev_emb.data_ptr(), prev_labels.data_ptr(), cur_emb.data_ptr()
        d.dis_bias = dis_bias.data_ptr()
        d.right_bits, d.wrong_bits, d.fg_rows, d.obj_rows = prep.right_bits.data_ptr(), prep.wrong_bits.data_ptr(), prep.fg_rows.data_ptr(), prep.obj_rows.data_ptr()
        d.counts, d.obj_offsets = prep.counts.data_ptr(), prep.obj_offsets.data_ptr()
        d.proxy_table, d.proxy_sqnorm = table.data_ptr(), sqn.data_ptr()
        d.prep_ready = prep_event.cuda_event if prep_event is not None else None
        d.proxies_ready = done_event.cuda_event if done_event is not None else None
        d.feat, d.head = feat.data_ptr(), head.data_ptr()
        for i in range(6):                      # measurement only: raw hipEvent_t handles (bench.py)
            d.probe[i] = probes[i] if probes is not None else None
        _lib.check(_lib.lib().aoc_frame_enqueue(ctypes.byref(d), ctypes.byref(self.state), _p(self.ws), self.ws.numel(), _stream()), "aoc_frame_enqueue")
        return feat, head


# ------------------------------------------------------------------------------------------ calibration side
def fg2bg_min(dis, n_obj, out=None, dis_obj_stride=None, out_obj_stride=None, n_ch=None, inner=None):
    """dis [O, c, ...] -> [O, 1, ...]: min over the other objects and over dim 1 (AEM:18-20).
    With explicit strides / sizes it reads and writes channel slices of a larger buffer in place."""
    _need_gpu(dis)
    if out is None:
        dis = _f32c(dis)
        n_ch = dis.shape[1]
        inner = dis.numel() // (n_obj * n_ch)
        out = torch.empty((n_obj, 1) + tuple(dis.shape[2:]), dtype=torch.float32, device=dis.device)
        dis_obj_stride, out_obj_stride = n_ch * inner, inner
    _lib.check(_lib.lib().aoc_fg2bg_min(_p(dis), n_obj, int(n_ch), int(inner), int(dis_obj_stride), _p(out), int(out_obj_stride), _stream()),
               "aoc_fg2bg_min")
    return out


def label_mix(labels_flat, rows):
    """out[p,:] = sum_o labels[p,o] * rows[o,:]   (aocnet.py:325)."""
    labels_flat, rows = _f32c(labels_flat), _f32c(rows)
    _need_gpu(labels_flat, rows)
    n, n_obj = labels_flat.shape
    C = rows.shape[1]
    out = torch.empty(n, C, dtype=torch.float32, device=rows.device)
    _lib.check(_lib.lib().aoc_label_mix(_p(labels_flat), _p(rows), n, n_obj, C, _p(out), _stream()), "aoc_label_mix")
    return out


def masked_mean_pool(emb, labels, epsilon, pixel_major=False, out_pos=None, out_neg=None, out_pos_sqnorm=None):
    """emb [F, hw, C]; labels [F, O, hw] (or [F, hw, O] with pixel_major) -> (pos [O,C], neg [O,C])   (ATT:155-189)."""
    emb, labels = _f32c(emb), _f32c(labels)
    _need_gpu(emb, labels)
    F_, hw, C = emb.shape
    n_obj = labels.shape[2] if pixel_major else labels.shape[1]
    L = _lib.lib()
    pos = torch.empty(n_obj, C, dtype=torch.float32, device=emb.device) if out_pos is None else out_pos
    neg = torch.empty(n_obj, C, dtype=torch.float32, device=emb.device) if out_neg is None else out_neg
    ws = _ws(L.aoc_masked_mean_pool_workspace_bytes(F_, hw, n_obj, C), emb.device)
    _lib.check(L.aoc_masked_mean_pool(_p(emb), _p(labels), F_, hw, C, n_obj, int(bool(pixel_major)), float(epsilon), _p(pos), _p(neg),
                                      _p(out_pos_sqnorm), _p(ws), ws.numel(), _stream()), "aoc_masked_mean_pool")
    return pos, neg


def film_gain(head, weight, bias):
    head, weight = _f32c(head), _f32c(weight)
    bias = _f32c(bias) if bias is not None else None
    _need_gpu(head, weight, bias)
    n_obj, D = head.shape
    channels = weight.shape[0]
    gain = torch.empty(n_obj, channels, dtype=torch.float32, device=head.device)
    _lib.check(_lib.lib().aoc_film_gain(_p(head), _p(weight), _p(bias), n_obj, D, channels, _p(gain), _stream()), "aoc_film_gain")
    return gain


def channel_scale(x, gain, out=None):
    """y[n,c,:,:] = gain[n,c] * x[n,c,:,:]."""
    x, gain = _f32c(x), _f32c(gain)
    _need_gpu(x, gain)
    planes = x.shape[0] * x.shape[1]
    hw = x.numel() // planes
    assert gain.numel() == planes
    y = torch.empty_like(x) if out is None else out
    _lib.check(_lib.lib().aoc_channel_scale(_p(x), _p(gain), planes, hw, _p(y), _stream()), "aoc_channel_scale")
    return y


def film_scale(x, head, weight, bias, out=None):
    """y[o,c,:,:] = (1 + tanh(head[o,:] . weight[c,:] + bias[c])) * x[o,c,:,:] in one launch (ATT:12-17, CLB:81-84)."""
    x, head, weight = _f32c(x), _f32c(head), _f32c(weight)
    bias = _f32c(bias) if bias is not None else None
    _need_gpu(x, head, weight, bias)
    n_obj, channels = x.shape[0], x.shape[1]
    assert head.shape[0] == n_obj and weight.shape[0] == channels and weight.shape[1] == head.shape[1]
    hw = x.numel() // (n_obj * channels)
    y = torch.empty_like(x) if out is None else out
    _lib.check(_lib.lib().aoc_film_scale(_p(x), _p(head), _p(weight), _p(bias), n_obj, head.shape[1], channels, hw, _p(y), _stream()), "aoc_film_scale")
    return y


def cond_gate_pool(z, phi_w, phi_b, k_rank, want_debug=False, want_plane_mean=False):
    """CL:23-43 -> gap [N, C] (optionally also scores [N,HW] and threshold [N]; with want_plane_mean also the plane means [N, C] of z,
    CLB:68, from the same pass that computes the scores)."""
    z, phi_w, phi_b = _f32c(z), _f32c(phi_w).reshape(-1), _f32c(phi_b).reshape(-1)
    _need_gpu(z, phi_w, phi_b)
    N, C = z.shape[0], z.shape[1]
    hw = z.numel() // (N * C)
    L = _lib.lib()
    gap = torch.empty(N, C, dtype=torch.float32, device=z.device)
    pm = torch.empty(N, C, dtype=torch.float32, device=z.device) if want_plane_mean else None
    scores = torch.empty(N, hw, dtype=torch.float32, device=z.device) if want_debug else None
    thr = torch.empty(N, dtype=torch.float32, device=z.device) if want_debug else None
    ws = _ws(L.aoc_cond_gate_pool_workspace_bytes(N, C, hw), z.device)
    _lib.check(L.aoc_cond_gate_pool_ex(_p(z), N, C, hw, _p(phi_w), _p(phi_b), int(k_rank), _p(gap), _p(pm), _p(scores), _p(thr), _p(ws), ws.numel(),
                                       _stream()), "aoc_cond_gate_pool_ex")
    out = (gap, scores, thr) if want_debug else gap
    if want_plane_mean:
        return (out + (pm,)) if want_debug else (gap, pm)
    return out


def linear(x, weight, bias):
    x, weight = _f32c(x), _f32c(weight)
    bias = _f32c(bias) if bias is not None else None
    _need_gpu(x, weight, bias)
    N, in_dim = x.shape
    out_dim = weight.shape[0]
    y = torch.empty(N, out_dim, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().aoc_linear(_p(x), _p(weight), _p(bias), N, in_dim, out_dim, _p(y), _stream()), "aoc_linear")
    return y


def plane_mean(x):
    """[N, C, H, W] -> [N, C] (CLB:68 avg_pool2d over the whole map)."""
    x = _f32c(x)
    _need_gpu(x)
    planes = x.shape[0] * x.shape[1]
    hw = x.numel() // planes
    out = torch.empty(x.shape[0], x.shape[1], dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().aoc_plane_mean(_p(x), planes, hw, _p(out), _stream()), "aoc_plane_mean")
    return out


def head_delta(head, plane_means):
    """torch.cat([head, plane_means.sum(0, keepdim=True) - plane_means], 1) in one launch (aoc_head_delta; decoding_module.py:126-130)."""
    head, plane_means = _f32c(head), _f32c(plane_means)
    _need_gpu(head, plane_means)
    n_obj, D = head.shape
    C = plane_means.shape[1]
    out = torch.empty(n_obj, D + C, dtype=torch.float32, device=head.device)
    _lib.check(_lib.lib().aoc_head_delta(_p(head), D, _p(plane_means), n_obj, C, _p(out), _stream()), "aoc_head_delta")
    return out


# ------------------------------------------------------------------------------------------ eval-loop memory policy
def confident_labels(probs_flat, exist_bits, join_label=None, unc_ratio=1.0):
    """aoc_confident_labels: probs [n_ch, n] -> (labels [n], confident [n] with 125 = uncertain, entropy [n])."""
    probs_flat = _f32c(probs_flat)
    _need_gpu(probs_flat, join_label)
    n_ch, n = probs_flat.shape
    dev = probs_flat.device
    labels = torch.empty(n, dtype=torch.int32, device=dev)
    confident = torch.empty(n, dtype=torch.int32, device=dev)
    entropy = torch.empty(n, dtype=torch.float32, device=dev)
    if join_label is not None:
        join_label = join_label.to(torch.int32).contiguous().reshape(-1)
        assert join_label.numel() == n
    _lib.check(_lib.lib().aoc_confident_labels(_p(probs_flat), n_ch, n, int(exist_bits) & 0xFFFFFFFF, _p(join_label), float(unc_ratio), _p(labels),
                                               _p(confident), _p(entropy), _stream()), "aoc_confident_labels")
    return labels, confident, entropy


def label_onehot_nearest(label_hw, h, w, n_obj):
    """aoc_label_onehot_nearest: int label map [H, W] -> float one-hot [h, w, n_obj] at matching resolution."""
    _need_gpu(label_hw)
    label_hw = label_hw.to(torch.int32).contiguous()
    H, W = label_hw.shape
    out = torch.empty(h, w, n_obj, dtype=torch.float32, device=label_hw.device)
    _lib.check(_lib.lib().aoc_label_onehot_nearest(_p(label_hw), H, W, int(h), int(w), int(n_obj), _p(out), _stream()), "aoc_label_onehot_nearest")
    return out


class MaskJF:
    """Device-side accumulator of the DAVIS region similarity J and boundary measure F (aoc_mask_jf_accumulate): add(pred, gt) enqueues
    three small kernels and never synchronises; totals() reads the four float64 accumulators back once."""

    def __init__(self, device):
        self.accum = torch.zeros(4, dtype=torch.float64, device=device)
        self.ws = None
        self.shape = None

    def add(self, pred, gt, n_obj):
        _need_gpu(pred, gt)
        pred, gt = pred.to(torch.int32).contiguous(), gt.to(torch.int32).contiguous()
        H, W = pred.shape
        assert gt.shape == (H, W)
        L = _lib.lib()
        clean = 1
        if self.shape != (H, W):
            self.ws = torch.zeros(int(L.aoc_mask_jf_workspace_bytes(H, W)), dtype=torch.uint8, device=pred.device)
            self.shape = (H, W)
        bound_pix = int(np.ceil(0.008 * np.hypot(H, W)))
        _lib.check(L.aoc_mask_jf_accumulate(_p(pred), _p(gt), H, W, int(n_obj), bound_pix, _p(self.ws), self.ws.numel(), clean, _p(self.accum), _stream()),
                   "aoc_mask_jf_accumulate")

    def totals(self):
        sj, sf, n, frames = self.accum.tolist()
        return dict(sum_j=sj, sum_f=sf, objects=n, frames=frames)


# ------------------------------------------------------------------------------------------ decoder-side streams (8f-4)
def plane_reduce(x, mode):
    """x [N, C, H, W] -> [N, C] plane sums of x (mode 0), x^2 (1) or |x| (2)."""
    x = _f32c(x)
    _need_gpu(x)
    N, C = x.shape[0], x.shape[1]
    hw = x.numel() // (N * C)
    out = torch.empty(N, C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().aoc_plane_reduce(_p(x), N * C, hw, int(mode), _p(out), _stream()), "aoc_plane_reduce")
    return out


def groupnorm_relu(x, groups, gamma, beta, eps=1e-5, residual=None, relu=True, out=None):
    """aoc_groupnorm_relu: [relu](GroupNorm(x) [+ residual]) in two streams over x (gct.py:69-90)."""
    x = _f32c(x)
    residual = _f32c(residual) if residual is not None else None
    _need_gpu(x, gamma, beta, residual)
    N, C = x.shape[0], x.shape[1]
    hw = x.numel() // (N * C)
    L = _lib.lib()
    y = torch.empty_like(x) if out is None else out
    ws = _ws(L.aoc_groupnorm_relu_workspace_bytes(N, int(groups)), x.device)
    # converted copies stay bound until the launch is enqueued: a temporary freed right after data_ptr() can be handed out again by the allocator
    g = _f32c(gamma) if gamma is not None else None
    b = _f32c(beta) if beta is not None else None
    _lib.check(L.aoc_groupnorm_relu(_p(x), N, C, hw, int(groups), _p(g), _p(b),
                                    float(eps), _p(residual), int(bool(relu)), _p(y), _p(ws), ws.numel(), _stream()), "aoc_groupnorm_relu")
    return y


def gct_gate(plane_sums, alpha, gamma, beta, eps, l1_mode=False):
    plane_sums = _f32c(plane_sums)
    _need_gpu(plane_sums, alpha, gamma, beta)
    N, C = plane_sums.shape
    gate = torch.empty(N, C, dtype=torch.float32, device=plane_sums.device)
    al, ga, be = _f32c(alpha).reshape(-1), _f32c(gamma).reshape(-1), _f32c(beta).reshape(-1)      # kept alive across the launch
    _lib.check(_lib.lib().aoc_gct_gate(_p(plane_sums), _p(al), _p(ga), _p(be), N, C,
                                       float(eps), int(bool(l1_mode)), _p(gate), _stream()), "aoc_gct_gate")
    return gate


def object_logit(x, weight_bias):
    """x [N, C, H, W], weight_bias [N, C + 1] (weights then bias, decoding_module.py:154-156) -> [N, 1, H, W]."""
    x = _f32c(x)
    weight_bias = _f32c(weight_bias)
    _need_gpu(x, weight_bias)
    N, C, H, W = x.shape
    assert weight_bias.shape == (N, C + 1)
    out = torch.empty(N, 1, H, W, dtype=torch.float32, device=x.device)
    base = weight_bias.data_ptr()
    _lib.check(_lib.lib().aoc_object_logit(_p(x), N, C, H * W, ctypes.c_void_p(base), C + 1, ctypes.c_void_p(base + 4 * C), C + 1, _p(out), _stream()),
               "aoc_object_logit")
    return out


def cond_codes(gap, plane_means, head, w1, b1, w2, b2, w3, b3):
    """aoc_cond_codes: the three conditioning codes of CLB:68-80 concatenated -> [N, 2C + D]."""
    gap, plane_means, head = _f32c(gap), _f32c(plane_means), _f32c(head)
    _need_gpu(gap, plane_means, head, w1, w2, w3)
    N, C = gap.shape
    D = head.shape[1]
    code = torch.empty(N, 2 * C + D, dtype=torch.float32, device=gap.device)
    _lib.check(_lib.lib().aoc_cond_codes(_p(gap), _p(plane_means), _p(head), _p(_f32c(w1)), _p(_f32c(b1)), _p(_f32c(w2)), _p(_f32c(b2)), _p(_f32c(w3)),
                                         _p(_f32c(b3)), N, C, D, _p(code), _stream()), "aoc_cond_codes")
    return code


def prehead(feat, weight, bias, n_groups, gamma, beta, eps, emb_hwc=None):
    """aoc_prehead: feat [O, n_in, h, w] -> [O, C + n_out, h, w] = (embedding expanded over the objects || ReLU(GN(conv1x1(feat))))."""
    feat = _f32c(feat)
    _need_gpu(feat, weight, bias, gamma, beta, emb_hwc)
    O, n_in, h, w = feat.shape
    weight = _f32c(weight).reshape(weight.shape[0], -1)
    n_out = weight.shape[0]
    assert weight.shape[1] == n_in, "DynamicPreHead runs with kernel_size = 1"
    C = 0
    if emb_hwc is not None:
        emb_hwc = _f32c(emb_hwc)
        C = emb_hwc.shape[-1]
    L = _lib.lib()
    out = torch.empty(O, C + n_out, h, w, dtype=torch.float32, device=feat.device)
    ws = _ws(L.aoc_prehead_workspace_bytes(O, n_out, n_out // n_groups, h * w), feat.device)
    bi, g, b = _f32c(bias), _f32c(gamma), _f32c(beta)            # kept alive across the launch (see groupnorm_relu)
    _lib.check(L.aoc_prehead(_p(feat), O, n_in, h * w, _p(weight), _p(bi), n_out, int(n_groups), _p(g), _p(b), float(eps),
                             _p(emb_hwc), C, _p(out), _p(ws), ws.numel(), _stream()), "aoc_prehead")
    return out
