"""Drop-in mirror of the reference matching library (networks/layers/matching.py ==
AOC-Net/adaptive_embedding_for_matching.py, "AEM") on the HIP library.

Same function names, positional order, defaults, output shapes and special cases as the
reference (cited per function); every computation runs in csrc/libaoc_hip.so.  ``n_chunks`` and
``allow_parallel`` are accepted and ignored: the fused kernels never materialise the
[m, O, n] / unfold tensors those flags exist to bound.

``use_float16=True`` (the default ARGUMENT of every reference function; the model passes MODEL_FLOAT16_MATCHING=False):
* the cluster path returns the constant the reference degrades to (scipy's kmeans2 rejects float16 -> bare ``except`` -> 5e4
  padding, AEM:275-286 -> feature 1.0);
* the dense, k = 1 proxy and local paths run the ``.half()`` arithmetic of the reference on the device (operands, norms, dot
  products and distances rounded to float16 where its float16 tensors round them; outputs fp32) -- see include/aoc_hip.h.
"""
import numpy as np
import torch

from . import ops

WRONG_LABEL_PADDING_DISTANCE = 5e4   # AEM:25
DEFAULT_CLUSTER_NUM = 16             # AEM:232
KMEANS_ITERS = 20                    # AEM:276


# ------------------------------------------------------------------------------------------ helpers
def _bias_vec(dis_bias, obj_nums, device):
    """dis_bias arrives as an nn.Parameter slice shaped [O,1,1,1] (aocnet.py:144) or a float."""
    if torch.is_tensor(dis_bias):
        b = dis_bias.detach().to(device=device, dtype=torch.float32).reshape(-1)
        if b.numel() == 1 and obj_nums > 1:
            b = b.expand(obj_nums)
        return b.contiguous()
    return torch.full((obj_nums,), float(dis_bias), dtype=torch.float32, device=device)


def _off_grid(h, w, rate, device):
    """[h, w, 1] bool: pixels that are NOT on the rate-strided grid (rows and columns 0, rate, 2 rate, ...)."""
    on = (torch.arange(h, device=device) % rate == 0)[:, None] & (torch.arange(w, device=device) % rate == 0)[None, :]
    return (~on)[:, :, None]


def _keep_big_objects_on_grid(labels, off_grid, rate, obj_pixel_num):
    """An object with more than obj_pixel_num * rate^2 labelled pixels in the frame loses its label off the strided grid (AEM:531-541 / 437-446)."""
    large = labels.sum(dim=(0, 1)) > obj_pixel_num * rate * rate
    return labels.masked_fill(off_grid & large[None, None, :], 0.0)


def _flatten_pool(all_ref_emb, all_ref_labels, h, w, atrous_rate, atrous_obj_pixel_num):
    """The reference pool as rows (AEM:507-579 / 715-787): every reference frame's pixels, one after the other.
    atrous_rate > 1, atrous_obj_pixel_num <= 0: only the pixels of the rate-strided grid are rows of the pool -- a device gather
        (aoc_atrous_subsample) of embeddings and labels;
    atrous_rate > 1, atrous_obj_pixel_num > 0: all pixels stay rows, but an object with more than atrous_obj_pixel_num * rate^2
        pixels in a frame keeps its label on the strided grid only.
    Never writes into the caller's labels (the reference does)."""
    C, O = all_ref_emb[0].size(2), all_ref_labels[0].size(2)
    rate = int(atrous_rate)
    embs, labs = [], []
    off_grid = None
    for e, l in zip(all_ref_emb, all_ref_labels):
        e, l = e.float(), l.float()
        if rate > 1 and atrous_obj_pixel_num > 0:
            if off_grid is None:
                off_grid = _off_grid(h, w, rate, e.device)
            l = _keep_big_objects_on_grid(l, off_grid, rate, atrous_obj_pixel_num)
        elif rate > 1:
            e, l = ops.atrous_subsample(e, rate), ops.atrous_subsample(l, rate)
        embs.append(e.reshape(-1, C))
        labs.append(l.reshape(-1, O))
    pool = embs[0] if len(embs) == 1 else torch.cat(embs, 0)
    labels = labs[0] if len(labs) == 1 else torch.cat(labs, 0)
    return pool.contiguous(), labels.contiguous()


def _emit(feature_planes, h, w, n_planes_per_obj, obj_nums, ori_size):
    """feature_planes [O * F, h, w] -> [1, H, W, O, F]; bilinear(align_corners) when ori_size differs
    (AEM:604-607).  An identity-size resize is an exact copy, so it doubles as the layout change."""
    H, W = (h, w) if ori_size is None else (int(ori_size[0]), int(ori_size[1]))
    out = torch.empty(1, H, W, obj_nums, n_planes_per_obj, dtype=torch.float32, device=feature_planes.device)
    ops.resize_bilinear_planes(feature_planes, H, W, out, 1, obj_nums * n_planes_per_obj)
    return out


# ------------------------------------------------------------------------------------------ cluster path (a2-a5)
def _level_list(cluster_num):
    multi = isinstance(cluster_num, (list, tuple))
    levels = [int(k) for k in cluster_num] if multi else [int(cluster_num)]
    if not levels or min(levels) < 1:
        raise ValueError("cluster_num must be a positive integer or a non-empty sequence of them")
    return levels, multi


def cluster_proxies(pool, labels_flat, cluster_num=DEFAULT_CLUSTER_NUM, init_rows=None, rng=None, iters=KMEANS_ITERS):
    """Adaptive-proxy construction, AEM:252-286, on the device.

    ``cluster_num``: the reference's ``cluster_num`` argument (AEM:231-232), or a sequence of them = the multi-level
    configuration (BASELINE.json configs[2], K in {8, 16, 32}): the reference function run once per level, in order; here all
    levels x objects advance as segments of ONE k-means chain (kmax = the largest level).

    Returns None when no row is labelled (AEM:588-589) or a dict with
      prep, levels, seg_k / init_rows (host lists per object; per level then per object for a sequence), seg_offsets (device,
      [L*O+1]: where (level, object)'s packed labels start), centroids [L*O,kmax,C], labels (packed), cluster_counts [L*O,kmax],
      proxies [L*O,2,kmax,C], proxy_sqnorm [L*O,2,kmax], counts (host).
    The only host round trip is the read-back of the O+1 row counts, which the reference's control
    flow needs anyway: sticky ``K_i = min(K_{i-1}, n_i)`` (AEM:268) and the rows scipy's
    ``minit='points'`` draws from numpy's global RandomState (``permutation(n_i)[:K_i]``).
    """
    levels, multi = _level_list(cluster_num)
    prep = ops.label_prep(labels_flat)
    n_obj = prep.n_obj
    counts = prep.counts.cpu().numpy()
    if int(counts[n_obj]) == 0:
        return None
    rng = np.random if rng is None else rng
    L, kmax = len(levels), max(levels)
    rows_host = np.zeros((L * n_obj, kmax), np.int32)
    seg_k, drawn = [], []
    for li, k in enumerate(levels):
        seg_k.append([])
        drawn.append([])
        given = None if init_rows is None else (init_rows[li] if multi else init_rows)
        for i in range(n_obj):
            k = min(k, int(counts[i]))                   # AEM:268 (sticky, per call = per level)
            seg_k[li].append(k)
            if k == 0:
                drawn[li].append(None)
                continue
            if given is not None and given[i] is not None:
                r = np.asarray(given[i], np.int64)[:k]
            else:
                r = np.asarray(rng.permutation(int(counts[i]))[:k], np.int64)   # scipy vq.py:519 on RandomState
            rows_host[li * n_obj + i, :k] = r
            drawn[li].append(r)
    dev = pool.device
    cap = prep.obj_rows.numel()
    init_dev = torch.from_numpy(rows_host).to(dev)
    rows, offs, seg_k_dev = ops.kmeans_replicate_levels(prep.obj_rows, prep.obj_offsets, n_obj, L, levels, rows_capacity=cap)
    centroids, labels, ccounts = ops.kmeans_segmented(pool, rows, offs, seg_k_dev, init_dev, kmax, iters, rows_capacity=L * cap, n_rep=L)
    proxies, sqnorm = ops.build_proxies(pool, prep.fg_rows, offs, seg_k_dev, labels, centroids)
    return dict(prep=prep, levels=levels, seg_k=seg_k if multi else seg_k[0], init_rows=drawn if multi else drawn[0], seg_offsets=offs,
                centroids=centroids, labels=labels, cluster_counts=ccounts, proxies=proxies, proxy_sqnorm=sqnorm, counts=counts)


def global_matching_for_eval_cluster(all_reference_embeddings, query_embeddings, all_reference_labels,
                                     n_chunks=20, dis_bias=0., ori_size=None, atrous_rate=1, use_float16=True,
                                     atrous_obj_pixel_num=0, init_rows=None, cluster_num=DEFAULT_CLUSTER_NUM):
    """AEM:480-613.  -> [1, H, W, O, 2]; [1, h, w, O, 1] of ones when no reference pixel is labelled.
    ``cluster_num`` (AEM:232) may be a sequence of levels -> [1, H, W, O, 2 * levels].  Every level follows AEM:231-332 semantics (20 Lloyd
    iterations, sticky K, failure -> constant), i.e. the function the model calls once per level; NOT matching.py:1711
    ``global_matching_for_eval_cluster2`` (iter=50, per-object K without the sticky rule, no try/except), whose per-level outputs differ."""
    ops.inference_only("global_matching_for_eval_cluster", query_embeddings, dis_bias, *all_reference_embeddings)
    h, w, embedding_dim = query_embeddings.size()
    obj_nums = all_reference_labels[0].size(2)
    dev = query_embeddings.device
    levels, _ = _level_list(cluster_num)
    L = len(levels)
    pool, labels_flat = _flatten_pool(all_reference_embeddings, all_reference_labels, h, w, atrous_rate, atrous_obj_pixel_num)
    if use_float16:
        # kmeans2 raises TypeError on float16 -> except -> constant 5e4 -> (sigmoid(5e4+b)-.5)*2 == 1
        right, _ = ops.label_bits(labels_flat, want_wrong=False)
        if not bool((right < 0).any()):                                   # bit 31 set <=> row kept
            return torch.ones(1, h, w, obj_nums, 1, device=dev)
        H, W = (h, w) if ori_size is None else ori_size
        return torch.ones(1, H, W, obj_nums, 2 * L, device=dev)
    cp = cluster_proxies(pool, labels_flat, cluster_num, init_rows)
    if cp is None:
        return torch.ones(1, h, w, obj_nums, 1, device=dev)               # AEM:588-589
    kmax = cp["proxies"].shape[2]
    query_flat = query_embeddings.reshape(-1, embedding_dim)
    bias = _bias_vec(dis_bias, obj_nums, dev)
    planes = torch.empty(obj_nums * 2 * L, h, w, dtype=torch.float32, device=dev)
    # set (level l, object o, centroid | centroid_avg f) -> plane o * 2L + 2l + f   (the concatenation order of AEM:599-612, per level)
    begin, size, off, sbias = [], [], [], []
    for l, k in enumerate(levels):
        for o in range(obj_nums):
            for f in range(2):
                begin.append(((l * obj_nums + o) * 2 + f) * kmax)
                size.append(min(k, kmax))
                off.append((o * 2 * L + 2 * l + f) * h * w)
                sbias.append(o)
    ops.proxy_corr_min(query_flat, cp["proxies"].reshape(-1, embedding_dim), cp["proxy_sqnorm"].reshape(-1), begin, size, off,
                       bias[torch.tensor(sbias, device=dev)], planes, 1, True)
    return _emit(planes, h, w, 2 * L, obj_nums, ori_size)


def _train_twin_labels(reference_labels, h, w, atrous_rate, atrous_obj_pixel_num):
    """AEM:437-446 (== 368-377, 648-657): with atrous_rate > 1 the training twins keep the label of every "big" object (more than
    atrous_obj_pixel_num * rate^2 pixels -- every labelled object when that parameter is 0) on the rate-strided grid only.  Same mask as the
    evaluation pool's (_keep_big_objects_on_grid); returns a new tensor (the reference writes into the caller's)."""
    if atrous_rate <= 1:
        return reference_labels
    return _keep_big_objects_on_grid(reference_labels.float(), _off_grid(h, w, int(atrous_rate), reference_labels.device), atrous_rate, atrous_obj_pixel_num)


def global_matching_cluster(reference_embeddings, query_embeddings, reference_labels,
                            n_chunks=100, dis_bias=0., ori_size=None, atrous_rate=1, use_float16=True, atrous_obj_pixel_num=0):
    """AEM:405-478 / matching.py:1324-1405 (training twin; single reference frame): atrous label masking of big objects
    (AEM:437-446) and a TWO-channel nothing-labelled early-out (AEM:455-456).  Inference only (no autograd graph)."""
    assert reference_embeddings.size()[:2] == reference_labels.size()[:2]     # AEM:430
    h, w, _ = query_embeddings.size()
    obj_nums = reference_labels.size(2)
    labels = _train_twin_labels(reference_labels, h, w, atrous_rate, atrous_obj_pixel_num)
    out = global_matching_for_eval_cluster([reference_embeddings], query_embeddings, [labels], n_chunks, dis_bias, ori_size, 1, use_float16, 0)
    if out.shape[-1] == 1:
        return torch.ones(1, h, w, obj_nums, 2, device=query_embeddings.device)
    return out


global_matching_cluster2 = global_matching_cluster   # name imported by aocnet.py:6


# ------------------------------------------------------------------------------------------ dense path (a6)
def global_matching_for_eval(all_reference_embeddings, query_embeddings, all_reference_labels,
                             n_chunks=20, dis_bias=0., ori_size=None, atrous_rate=1, use_float16=True, atrous_obj_pixel_num=0):
    """AEM:688-817.  -> [1, H, W, O, 1]; ones when nothing is labelled (AEM:796-797).  No host sync."""
    ops.inference_only("global_matching_for_eval", query_embeddings, dis_bias, *all_reference_embeddings)
    h, w, embedding_dim = query_embeddings.size()
    obj_nums = all_reference_labels[0].size(2)
    dev = query_embeddings.device
    pool, labels_flat = _flatten_pool(all_reference_embeddings, all_reference_labels, h, w, atrous_rate, atrous_obj_pixel_num)
    prep = ops.label_prep(labels_flat)
    planes = torch.empty(obj_nums, h, w, dtype=torch.float32, device=dev)
    ops.dense_match(query_embeddings.reshape(-1, embedding_dim), pool, prep, _bias_vec(dis_bias, obj_nums, dev),
                    planes, 1, h * w, True, precision="f16" if use_float16 else None)     # AEM:801-803
    if ori_size is not None:
        # the all-unlabelled early-out of the reference keeps the map at (h, w) even with ori_size
        if int(prep.counts[obj_nums]) == 0:
            return torch.ones(1, h, w, obj_nums, 1, device=dev)
    return _emit(planes, h, w, 1, obj_nums, ori_size)


def global_matching(reference_embeddings, query_embeddings, reference_labels,
                    n_chunks=100, dis_bias=0., ori_size=None, atrous_rate=1, use_float16=True, atrous_obj_pixel_num=0):
    """AEM:616-685 (training twin).  Inference only (no autograd graph)."""
    assert reference_embeddings.size()[:2] == reference_labels.size()[:2]     # AEM:641
    h, w, _ = query_embeddings.size()
    labels = _train_twin_labels(reference_labels, h, w, atrous_rate, atrous_obj_pixel_num)      # AEM:648-657
    return global_matching_for_eval([reference_embeddings], query_embeddings, [labels], n_chunks,
                                    dis_bias, ori_size, 1, use_float16, 0)


# ------------------------------------------------------------------------------------------ k = 1 proxy path (a7)
def global_matching_for_eval_proxy(all_reference_embeddings, query_embeddings, all_reference_labels,
                                   n_chunks=20, dis_bias=0., ori_size=None, atrous_rate=1, use_float16=True, atrous_obj_pixel_num=0):
    """matching.py:2518-2662 (the AEM:819-873 copy references undefined names).  ``all_reference_embeddings``
    is the [O, C] tensor of mean-pooled proxies (aocnet.py:314-315); out[i,o] = d(q_i, proxy_o).  -> [1,H,W,O,1]"""
    ops.inference_only("global_matching_for_eval_proxy", query_embeddings, dis_bias, all_reference_embeddings)
    h, w, embedding_dim = query_embeddings.size()
    obj_nums = all_reference_labels[0].size(2)
    dev = query_embeddings.device
    if len(all_reference_labels) == 0:
        return torch.ones(1, h, w, obj_nums, 1, device=dev)
    proxies = all_reference_embeddings.float().contiguous()
    planes = torch.empty(obj_nums, h, w, dtype=torch.float32, device=dev)
    ops.proxy_corr_min(query_embeddings.reshape(-1, embedding_dim), proxies, None, list(range(obj_nums)), [1] * obj_nums,
                       [o * h * w for o in range(obj_nums)], _bias_vec(dis_bias, obj_nums, dev), planes, 1, True, float16=bool(use_float16))
    return _emit(planes, h, w, 1, obj_nums, ori_size)


def global_matching_proxy(reference_embeddings, query_embeddings, reference_labels,
                          n_chunks=100, dis_bias=0., ori_size=None, atrous_rate=1, use_float16=True, atrous_obj_pixel_num=0):
    """AEM:336-402 (training twin): ones when no reference pixel is labelled (AEM:382-386)."""
    h, w, _ = query_embeddings.size()
    obj_nums = reference_labels.size(2)
    reference_labels = _train_twin_labels(reference_labels, h, w, atrous_rate, atrous_obj_pixel_num)   # AEM:368-377
    right, _ = ops.label_bits(reference_labels.reshape(-1, obj_nums), want_wrong=False)
    if not bool((right < 0).any()):
        return torch.ones(1, h, w, obj_nums, 1, device=query_embeddings.device)
    return global_matching_for_eval_proxy(reference_embeddings, query_embeddings, [reference_labels], n_chunks,
                                          dis_bias, ori_size, 1, use_float16, 0)


# ------------------------------------------------------------------------------------------ local path (a8)
def local_matching(prev_frame_embedding, query_embedding, prev_frame_labels, dis_bias=0., multi_local_distance=[15],
                   ori_size=None, atrous_rate=1, use_float16=True, allow_downsample=True, allow_parallel=True):
    """AEM:968-1060.  -> [1, H, W, O, len(multi_local_distance)], channel order [max, d_0, d_1, ...]."""
    ops.inference_only("local_matching", prev_frame_embedding, query_embedding, dis_bias)
    h, w, _ = prev_frame_embedding.size()
    if ori_size is None:
        ori_size = (h, w)
    obj_num = prev_frame_labels.size(2)
    dev = query_embedding.device
    f16 = bool(use_float16)                                                # AEM:1002-1005
    radii = [int(r) for r in multi_local_distance]
    right, _ = ops.label_bits(prev_frame_labels.reshape(-1, obj_num), want_wrong=False)
    if allow_downsample:
        H, W = int(h / 2) + 1, int(w / 2) + 1                              # AEM:939
        q = ops.resize_bilinear_hwc(query_embedding, H, W, float16=f16)
        p = ops.resize_bilinear_hwc(prev_frame_embedding, H, W, float16=f16)
    else:
        H, W = h, w
        q, p = query_embedding, prev_frame_embedding
    if (H, W) != tuple(ori_size):
        # AEM:1017-1018: labels go to the matching resolution by 'nearest' FROM THEIR OWN size (h, w)
        right = ops.resize_nearest_bits(right, h, w, H, W)
    elif (H, W) != (h, w):
        raise ValueError("local_matching: label map and distance map sizes differ")   # reference would fail in unfold too
    feats = ops.local_window_match(q, p, right, radii, _bias_vec(dis_bias, obj_num, dev), obj_num, True, atrous_rate=int(atrous_rate),
                                   float16=f16)                                        # [O, nr, H, W]; AEM:949-959 window stride
    nr = len(radii)
    out = torch.empty(1, ori_size[0], ori_size[1], obj_num, nr, dtype=torch.float32, device=dev)
    ops.resize_bilinear_planes(feats.reshape(obj_num * nr, H, W), int(ori_size[0]), int(ori_size[1]), out, 1, obj_num * nr)
    return out


local_matching_proxy = local_matching   # AEM:1064-1156 is a verbatim copy of AEM:968-1060


# ------------------------------------------------------------------------------------------ fg -> bg (a9)
def foreground2background(dis, obj_num):
    """AEM:9-23: per object the min over all other objects (and over dim 1, which the reference
    concatenates on).  dis [O, c, ...] -> [O, 1, ...]."""
    if obj_num == 1:
        return dis
    ops.inference_only("foreground2background", dis)
    return ops.fg2bg_min(dis, obj_num)
