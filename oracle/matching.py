"""CPU restatement of the reference matching library (TEST INFRASTRUCTURE ONLY).

Reference file (AEM): /root/reference/AOC-Net/adaptive_embedding_for_matching.py
(== AOC-Net/complete_project/AOCNet/networks/layers/matching.py for these functions).
Every function cites the AEM lines it follows.  Differences from the reference are
structural only: no query chunking (chunking never changes a value), no Python loops
over chunks, scipy's kmeans2 replaced by the bit-exact C restatement in oracle.kmeans,
and the k-means initial rows may be passed explicitly (``init_rows``) instead of being
drawn from numpy's global RandomState (when omitted they are drawn exactly as scipy does).

All tensors are torch CPU float32.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import kmeans as _km

WRONG_LABEL_PADDING_DISTANCE = 5e4  # AEM:25
KMEANS_ITERS = 20                   # AEM:276
DEFAULT_CLUSTER_NUM = 16            # AEM:232


# --------------------------------------------------------------------------- a1
def pairwise_distances(x, x2, y, y2):
    """AEM:29-44.  d[i,j] = x2[i] + y2[j] - 2 x_i.y_j"""
    return x2.unsqueeze(1) + y2.unsqueeze(0) - 2.0 * torch.matmul(x, y.t())


def flattened_pairwise_distances(ref, ref_sq, query, query_sq):
    """AEM:47-59: note the argument swap -> result is [query, ref]."""
    return pairwise_distances(query, query_sq, ref, ref_sq)


def proto_transform(d, bias):
    """AEM:393/467/602/676/808/864/1049:  (sigmoid(d + bias) - 0.5) * 2"""
    return (torch.sigmoid(d + bias) - 0.5) * 2


# --------------------------------------------------------------------------- a9
def foreground2background(dis, obj_num):
    """AEM:9-23: for each object the elementwise min over all OTHER objects."""
    if obj_num == 1:
        return dis
    out = []
    for i in range(obj_num):
        others = torch.cat([dis[j].unsqueeze(0) for j in range(obj_num) if j != i], dim=1)
        out.append(torch.min(others, dim=1, keepdim=True)[0])
    return torch.cat(out, dim=0)


# ------------------------------------------------------ atrous / pool flattening
def _flatten_reference_pool(all_ref_emb, all_ref_labels, h, w, atrous_rate, atrous_obj_pixel_num):
    """AEM:507-579 (identical in AEM:715-787): concat the R reference frames, with the
    optional atrous subsampling of reference pixels.  NB the reference mutates the
    caller's label tensors in the ``atrous_obj_pixel_num > 0`` branch (AEM:526); the
    oracle works on clones."""
    embedding_dim = all_ref_emb[0].size(2)
    obj_nums = all_ref_labels[0].size(2)
    embs, labs = [], []
    if atrous_obj_pixel_num > 0:
        sel = None
        if atrous_rate > 1:
            h_pad = (atrous_rate - h % atrous_rate) % atrous_rate
            w_pad = (atrous_rate - w % atrous_rate) % atrous_rate
            sel = torch.zeros(h + h_pad, w + w_pad)
            sel = sel.view((h + h_pad) // atrous_rate, atrous_rate, (w + w_pad) // atrous_rate, atrous_rate)
            sel[:, 0, :, 0] = 1.
            sel = sel.reshape(h + h_pad, w + w_pad, 1)[:h, :w]
        for e, l in zip(all_ref_emb, all_ref_labels):
            l = l.clone()
            if atrous_rate > 1:
                big = l.sum(dim=(0, 1)) > (atrous_obj_pixel_num * atrous_rate ** 2)
                l[:, :, big] = l[:, :, big] * sel
            embs.append(e.reshape(-1, embedding_dim))
            labs.append(l.reshape(-1, obj_nums))
    else:
        for e, l in zip(all_ref_emb, all_ref_labels):
            if atrous_rate > 1:
                h_pad = (atrous_rate - h % atrous_rate) % atrous_rate
                w_pad = (atrous_rate - w % atrous_rate) % atrous_rate
                if h_pad > 0 or w_pad > 0:
                    e = F.pad(e, (0, 0, 0, w_pad, 0, h_pad))
                    l = F.pad(l, (0, 0, 0, w_pad, 0, h_pad))
                e = e.reshape((h + h_pad) // atrous_rate, atrous_rate, (w + w_pad) // atrous_rate, atrous_rate, -1)
                l = l.reshape((h + h_pad) // atrous_rate, atrous_rate, (w + w_pad) // atrous_rate, atrous_rate, -1)
                e = e[:, 0, :, 0, :].contiguous()
                l = l[:, 0, :, 0, :].contiguous()
            embs.append(e.reshape(-1, embedding_dim))
            labs.append(l.reshape(-1, obj_nums))
    return torch.cat(embs, 0), torch.cat(labs, 0)


def _keep_foreground_rows(ref_flat, labels_flat):
    """AEM:585-591 / 793-799: keep rows whose label sum exceeds 0.9."""
    keep = labels_flat.sum(dim=1) > 0.9
    return ref_flat[keep], labels_flat[keep], keep


def _finish(nn_features, h, w, obj_nums, dis_bias, ori_size):
    """AEM:601-607 / 807-813: reshape, proto-mask transform, optional bilinear resize."""
    x = nn_features.reshape(1, h, w, obj_nums, 1)
    x = proto_transform(x, dis_bias.reshape(1, 1, 1, -1, 1))
    if ori_size is not None:
        x = x.reshape(h, w, obj_nums, 1).permute(2, 3, 0, 1)
        x = F.interpolate(x, size=ori_size, mode='bilinear', align_corners=True)
        x = x.permute(2, 3, 0, 1).reshape(1, ori_size[0], ori_size[1], obj_nums, 1)
    return x


def _as_bias(dis_bias, obj_nums):
    if not torch.is_tensor(dis_bias):
        dis_bias = torch.full((obj_nums,), float(dis_bias))
    return dis_bias.detach().float().reshape(-1)


# --------------------------------------------------------------------------- a6
def nearest_neighbor_features_per_object(ref_flat, query_flat, labels_flat):
    """AEM:178-227 + 61-89 (dense pixel-level matching, de-chunked):
    out[i,o] = min_j ( d(q_i, r_j) + 5e4 * (label[j,o] < 0.1) )    -> [m, O, 1]
    float16 operands (use_float16=True, AEM:801-803) run the SAME tensor operations on float16 tensors: the wrong-label mask is cast
    to the operands' dtype (AEM:65-66), so norms, dot products, distances, the padded sum and the min are float16."""
    wrong = (labels_flat < 0.1).permute(1, 0).to(ref_flat.dtype)          # AEM:197-198, 65-68
    ref_sq = ref_flat.pow(2).sum(1)                                       # AEM:199
    query_sq = query_flat.pow(2).sum(1)                                   # AEM:200
    d = flattened_pairwise_distances(ref_flat, ref_sq, query_flat, query_sq)  # [m, n]
    out = []
    for o in range(wrong.size(0)):     # loop only to bound memory; same values as AEM:84-88
        out.append((d + wrong[o].unsqueeze(0) * WRONG_LABEL_PADDING_DISTANCE).min(dim=1, keepdim=True)[0])
    return torch.stack(out, dim=1)                                        # [m, O, 1]


def global_matching_for_eval(all_reference_embeddings, query_embeddings, all_reference_labels,
                             n_chunks=20, dis_bias=0., ori_size=None, atrous_rate=1,
                             use_float16=False, atrous_obj_pixel_num=0):
    """AEM:688-817.  -> [1, h, w, O, 1].  use_float16: the operands are cast with .half() after the row filter (AEM:801-803); the
    bias is added and the sigmoid taken in fp32 by type promotion (AEM:808), the result is fp32 (AEM:815-816)."""
    h, w, embedding_dim = query_embeddings.size()
    obj_nums = all_reference_labels[0].size(2)
    ref_flat, labels_flat = _flatten_reference_pool(all_reference_embeddings, all_reference_labels,
                                                    h, w, atrous_rate, atrous_obj_pixel_num)
    query_flat = query_embeddings.reshape(-1, embedding_dim)
    ref_flat, labels_flat, _ = _keep_foreground_rows(ref_flat, labels_flat)
    if labels_flat.size(0) == 0:
        return torch.ones(1, h, w, obj_nums, 1)                           # AEM:796-797
    if use_float16:
        ref_flat, query_flat = ref_flat.half(), query_flat.half()         # AEM:801-803
    nn = nearest_neighbor_features_per_object(ref_flat, query_flat, labels_flat)
    return _finish(nn, h, w, obj_nums, _as_bias(dis_bias, obj_nums), ori_size).float()


def _train_twin_labels(reference_labels, h, w, atrous_rate, atrous_obj_pixel_num):
    """AEM:437-446 (== AEM:368-377, 648-657): the training twins mask the labels of every "big" object with the
    atrous grid whenever atrous_rate > 1 (the reference writes into the caller's tensor; the oracle works on a clone)."""
    if atrous_rate <= 1:
        return reference_labels
    h_pad = (atrous_rate - h % atrous_rate) % atrous_rate
    w_pad = (atrous_rate - w % atrous_rate) % atrous_rate
    sel = torch.zeros(h + h_pad, w + w_pad)
    sel = sel.view((h + h_pad) // atrous_rate, atrous_rate, (w + w_pad) // atrous_rate, atrous_rate)
    sel[:, 0, :, 0] = 1.
    sel = sel.reshape(h + h_pad, w + w_pad, 1)[:h, :w]
    labels = reference_labels.clone()
    big = labels.sum(dim=(0, 1)) > (atrous_obj_pixel_num * atrous_rate ** 2)
    labels[:, :, big] = labels[:, :, big] * sel
    return labels


def global_matching(reference_embeddings, query_embeddings, reference_labels,
                    n_chunks=100, dis_bias=0., ori_size=None, atrous_rate=1,
                    use_float16=False, atrous_obj_pixel_num=0):
    """AEM:616-685 (training twin, single reference frame)."""
    h, w, _ = query_embeddings.size()
    labels = _train_twin_labels(reference_labels, h, w, atrous_rate, atrous_obj_pixel_num)
    return global_matching_for_eval([reference_embeddings], query_embeddings, [labels],
                                    n_chunks, dis_bias, ori_size, 1, use_float16, 0)


# ------------------------------------------------------------------- a2, a3, a4
def build_adaptive_proxies(ref_flat, labels_flat, cluster_num=DEFAULT_CLUSTER_NUM, init_rows=None, rng=None):
    """AEM:252-286.  Per object (background included), in order:
      rows with label > 0.9 (AEM:252,263-264) -> K_i = min(K, n_i), STICKY across objects
      (AEM:268 overwrites the loop variable) -> kmeans2(X_i, K_i, 'points', iter=20) (AEM:276)
      -> centroid[K_i, C], label[n_i] -> centroid_avg over np.unique(label), computed by
      indexing the GLOBAL foreground array with object-local row numbers (AEM:280, reproduced
      as is) -> squared norms (AEM:282).  K_i == 0 or any exception -> None (AEM:271-273,283-286).

    Returns a list (one entry per object) of None or a dict with keys
    centroid, centroid_avg, labels, counts, init_rows, k.
    """
    ref_np = ref_flat.detach().cpu().numpy()
    right = (labels_flat > 0.9).permute(1, 0)
    proxies = []
    for i in range(right.size(0)):
        idx = torch.nonzero(right[i]).squeeze(1).numpy()
        x_i = ref_np[idx]
        cluster_num = min(cluster_num, x_i.shape[0])                      # AEM:268 (sticky)
        if cluster_num == 0:
            proxies.append(None)
            continue
        try:
            if init_rows is not None and init_rows[i] is not None:
                rows = np.asarray(init_rows[i], np.int64)[:cluster_num]
            else:
                rows = _km.draw_init_rows(x_i.shape[0], cluster_num, rng)
            centroid, label, counts = _km.kmeans2_matrix(x_i, x_i[rows], KMEANS_ITERS)
            uniq = np.unique(label)
            # AEM:280: index_select(reference_embeddings_flat, nonzero(label == j)) -- the
            # GLOBAL fg array at object-LOCAL indices; torch.sum over dim 0, / count.
            avg = torch.cat([(torch.sum(ref_flat[torch.from_numpy(np.nonzero(label == j)[0])], 0)
                              / float(np.sum(label == j))).unsqueeze(0) for j in uniq], 0)
            proxies.append(dict(centroid=torch.from_numpy(centroid), centroid_avg=avg,
                                labels=label, counts=counts, init_rows=rows, k=cluster_num))
        except Exception:                                                  # AEM:283-286
            proxies.append(None)
    return proxies


def nearest_neighbor_features_cluster(ref_flat, query_flat, labels_flat, cluster_num=DEFAULT_CLUSTER_NUM,
                                      init_rows=None, rng=None, return_proxies=False):
    """AEM:231-332 -> [features1, features2], each [m, O, 1]:
    min over the K k-means centroids / over the K' 'centroid_avg' proxies of the pairwise
    distance to every query pixel (AEM:92-110, 316-319); absent object -> constant 5e4."""
    m = query_flat.size(0)
    query_sq = query_flat.pow(2).sum(1)                                   # AEM:250
    proxies = build_adaptive_proxies(ref_flat, labels_flat, cluster_num, init_rows, rng)
    f1, f2 = [], []
    for p in proxies:
        if p is None:
            pad = torch.ones(m, 1, 1) * WRONG_LABEL_PADDING_DISTANCE      # AEM:312-313
            f1.append(pad)
            f2.append(pad.clone())
            continue
        for key, dst in (("centroid", f1), ("centroid_avg", f2)):
            c = p[key]
            d = flattened_pairwise_distances(c, c.pow(2).sum(1), query_flat, query_sq)
            dst.append(d.unsqueeze(1).min(dim=2, keepdim=True)[0])        # AEM:107-109
    out = [torch.cat(f1, 1), torch.cat(f2, 1)]
    return (out, proxies) if return_proxies else out


def global_matching_for_eval_cluster(all_reference_embeddings, query_embeddings, all_reference_labels,
                                     n_chunks=20, dis_bias=0., ori_size=None, atrous_rate=1,
                                     use_float16=False, atrous_obj_pixel_num=0,
                                     init_rows=None, rng=None, return_proxies=False, cluster_num=DEFAULT_CLUSTER_NUM):
    """AEM:480-613 (fp32 path) -> [1, h, w, O, 2]   (all-background early-out: [1,h,w,O,1] ones).

    ``cluster_num`` is the ``cluster_num`` argument of AEM:231-232 (default 16).  A sequence of values restates the
    multi-level configuration (BASELINE.json configs[2], K in {8, 16, 32}): the reference function is run once per
    level, in order, on the same inputs (numpy's global RandomState is consumed level by level) and the [.., 2]
    outputs are concatenated on the last axis -> [1, h, w, O, 2 * levels].  ``init_rows`` is then a list per level."""
    assert not use_float16, "oracle restates the fp32 path"
    if isinstance(cluster_num, (list, tuple)):
        outs, prox = [], []
        for li, k in enumerate(cluster_num):
            o, p = global_matching_for_eval_cluster(all_reference_embeddings, query_embeddings, all_reference_labels, n_chunks,
                                                    dis_bias, ori_size, atrous_rate, use_float16, atrous_obj_pixel_num,
                                                    None if init_rows is None else init_rows[li], rng, True, int(k))
            outs.append(o)
            prox.append(p)
        if outs[0].shape[-1] == 1:                                             # nothing labelled: every level early-outs
            return (outs[0], None) if return_proxies else outs[0]
        out = torch.cat(outs, 4)
        return (out, prox) if return_proxies else out
    h, w, embedding_dim = query_embeddings.size()
    obj_nums = all_reference_labels[0].size(2)
    ref_flat, labels_flat = _flatten_reference_pool(all_reference_embeddings, all_reference_labels,
                                                    h, w, atrous_rate, atrous_obj_pixel_num)
    query_flat = query_embeddings.reshape(-1, embedding_dim)
    ref_flat, labels_flat, _ = _keep_foreground_rows(ref_flat, labels_flat)
    if labels_flat.size(0) == 0:
        out = torch.ones(1, h, w, obj_nums, 1)                            # AEM:588-589
        return (out, None) if return_proxies else out
    feats, proxies = nearest_neighbor_features_cluster(ref_flat, query_flat, labels_flat,
                                                       int(cluster_num), init_rows, rng, True)
    bias = _as_bias(dis_bias, obj_nums)
    out = torch.cat([_finish(f, h, w, obj_nums, bias, ori_size) for f in feats], 4)   # AEM:599-612
    return (out, proxies) if return_proxies else out


# --------------------------------------------------------------------------- a7
def global_matching_for_eval_proxy(all_reference_embeddings, query_embeddings, all_reference_labels,
                                   n_chunks=20, dis_bias=0., ori_size=None, atrous_rate=1,
                                   use_float16=False, atrous_obj_pixel_num=0):
    """matching.py:2518-2662 (the runnable copy; AEM:819-873 has undefined names).
    ``all_reference_embeddings`` is the [O, C] tensor of k=1 proxies (aocnet.py:314-315):
    out[i,o] = d(q_i, proxy_o), no min.  -> [1, h, w, O, 1]
    use_float16=True raises UnboundLocalError in BOTH copies of the reference (matching.py:2640-2646 reads a name it never assigns);
    the oracle restates what the code evidently means -- the query and the proxies cast with .half() -- PARITY UNPINNED for that mode."""
    h, w, embedding_dim = query_embeddings.size()
    obj_nums = all_reference_labels[0].size(2)
    query_flat = query_embeddings.reshape(-1, embedding_dim)
    proxies = all_reference_embeddings
    if use_float16:
        query_flat, proxies = query_flat.half(), proxies.half()
    d = flattened_pairwise_distances(proxies, proxies.pow(2).sum(1), query_flat, query_flat.pow(2).sum(1))
    return _finish(d, h, w, obj_nums, _as_bias(dis_bias, obj_nums), ori_size).float()


def global_matching_proxy(reference_embeddings, query_embeddings, reference_labels,
                          n_chunks=100, dis_bias=0., ori_size=None, atrous_rate=1,
                          use_float16=False, atrous_obj_pixel_num=0):
    """AEM:336-402 (training twin; fp32).  Early-out to ones when no reference
    pixel is labelled (AEM:382-386)."""
    assert not use_float16
    h, w, embedding_dim = query_embeddings.size()
    obj_nums = reference_labels.size(2)
    labels_flat = _train_twin_labels(reference_labels, h, w, atrous_rate, atrous_obj_pixel_num).reshape(-1, obj_nums)
    if int((labels_flat.sum(1) > 0.9).sum()) == 0:
        return torch.ones(1, h, w, obj_nums, 1)
    return global_matching_for_eval_proxy(reference_embeddings, query_embeddings, [reference_labels],
                                          n_chunks, dis_bias, ori_size, 1, False, 0)


def global_matching_cluster(reference_embeddings, query_embeddings, reference_labels,
                            n_chunks=100, dis_bias=0., ori_size=None, atrous_rate=1,
                            use_float16=False, atrous_obj_pixel_num=0, init_rows=None, rng=None):
    """AEM:405-478 == matching.py:1324-1405 ``global_matching_cluster2`` (training twin of the adaptive-proxy
    matching; the AEM copy calls a name that only matching.py defines).  Differences from the eval form: labels of
    big objects are masked with the atrous grid (AEM:437-446) and the nothing-labelled early-out has TWO channels
    (AEM:455-456).  -> [1, h, w, O, 2]"""
    assert not use_float16
    h, w, _ = query_embeddings.size()
    obj_nums = reference_labels.size(2)
    labels = _train_twin_labels(reference_labels, h, w, atrous_rate, atrous_obj_pixel_num)
    if int((labels.reshape(-1, obj_nums).sum(1) > 0.9).sum()) == 0:
        return torch.ones(1, h, w, obj_nums, 2)
    return global_matching_for_eval_cluster([reference_embeddings], query_embeddings, [labels], n_chunks, dis_bias,
                                            ori_size, 1, False, 0, init_rows, rng)


# --------------------------------------------------------------------------- a8
def local_pairwise_distances(x, y, max_distance=9, atrous_rate=1, allow_downsample=True):
    """AEM:921-963 without the 1.6 GB unfold: d[y,x,(dy,dx)] = x2 + y2_pad - 2 x.y_pad where the
    pad of y is 0 and the pad of y2 is 5e4 (so out-of-image neighbours are 'far')."""
    ori_h, ori_w, _ = x.size()
    x = x.permute(2, 0, 1).unsqueeze(0)
    y = y.permute(2, 0, 1).unsqueeze(0)
    if allow_downsample:
        down = (int(ori_h / 2) + 1, int(ori_w / 2) + 1)                  # AEM:939
        x = F.interpolate(x, size=down, mode='bilinear', align_corners=True)
        y = F.interpolate(y, size=down, mode='bilinear', align_corners=True)
    _, c, height, width = x.size()
    x2 = x.pow(2).sum(1).reshape(height, width)
    y2 = y.pow(2).sum(1).reshape(1, 1, height, width)
    pad = max_distance - max_distance % atrous_rate                       # AEM:949
    py = F.pad(y, (pad, pad, pad, pad))
    py2 = F.pad(y2, (pad, pad, pad, pad), mode='constant', value=WRONG_LABEL_PADDING_DISTANCE)
    xs = x[0].permute(1, 2, 0)                                            # [H, W, C]
    n_off = 2 * pad // atrous_rate + 1
    out = torch.empty(height, width, n_off * n_off, dtype=x.dtype)
    for oy in range(n_off):
        for ox in range(n_off):
            ys = py[0, :, oy * atrous_rate: oy * atrous_rate + height, ox * atrous_rate: ox * atrous_rate + width]
            ys2 = py2[0, 0, oy * atrous_rate: oy * atrous_rate + height, ox * atrous_rate: ox * atrous_rate + width]
            dot = (xs.float() * ys.permute(1, 2, 0).float()).sum(2).to(x.dtype)      # matmul: exact products, fp32 accumulation, one rounding
            out[:, :, oy * n_off + ox] = x2 + ys2 - 2. * dot              # AEM:961
    return out


def local_matching(prev_frame_embedding, query_embedding, prev_frame_labels, dis_bias=0.,
                   multi_local_distance=(15,), ori_size=None, atrous_rate=1, use_float16=False,
                   allow_downsample=True, allow_parallel=True):
    """AEM:968-1060 (== local_matching_proxy AEM:1064-1156).  -> [1, h, w, O, len(multi_local_distance)]
    with channel order [max_distance, d_0, d_1, ...] (AEM:1034-1046)."""
    multi_local_distance = list(multi_local_distance)
    max_distance = multi_local_distance[-1]
    if use_float16:                                                        # AEM:1002-1005
        query_embedding, prev_frame_embedding = query_embedding.half(), prev_frame_embedding.half()
    if ori_size is None:
        ori_size = tuple(prev_frame_embedding.size()[:2])
    obj_num = prev_frame_labels.size(2)
    d = local_pairwise_distances(query_embedding, prev_frame_embedding, max_distance, atrous_rate, allow_downsample)
    height, width = d.size()[:2]
    labels = prev_frame_labels.permute(2, 0, 1).unsqueeze(1)
    if (height, width) != tuple(ori_size):
        labels = F.interpolate(labels, size=(height, width), mode='nearest')   # AEM:1017-1018
    pad = max_distance - max_distance % atrous_rate
    amax = pad // atrous_rate
    n_off = 2 * amax + 1
    pl = F.pad(labels, (pad, pad, pad, pad), mode='constant', value=0)
    masks = torch.empty(height, width, n_off * n_off, obj_num, dtype=torch.bool)
    for oy in range(n_off):
        for ox in range(n_off):
            sl = pl[:, 0, oy * atrous_rate: oy * atrous_rate + height, ox * atrous_rate: ox * atrous_rate + width]
            masks[:, :, oy * n_off + ox, :] = sl.permute(1, 2, 0) > 0.9   # AEM:1027-1028
    padv = torch.tensor(WRONG_LABEL_PADDING_DISTANCE, dtype=d.dtype)     # AEM:1001-1005 (pad.half() in float16 mode)
    d_masked = torch.where(masks, d.unsqueeze(-1).expand(-1, -1, -1, obj_num), padv)   # AEM:1032
    multi = [d_masked.min(dim=2)[0].permute(2, 0, 1).unsqueeze(1)]
    r = d_masked.reshape(height, width, n_off, n_off, obj_num)
    for ld in multi_local_distance[:-1]:
        ld = ld // atrous_rate
        s, e = amax - ld, amax + ld + 1
        sub = r[:, :, s:e, s:e, :].reshape(height, width, -1, obj_num)
        multi.append(sub.min(dim=2)[0].permute(2, 0, 1).unsqueeze(1))
    multi = torch.cat(multi, dim=1)
    bias = _as_bias(dis_bias, obj_num)
    multi = proto_transform(multi, bias.reshape(-1, 1, 1, 1)).float()     # AEM:1049-1052 (fp32 by type promotion, then .float())
    if (height, width) != tuple(ori_size):
        multi = F.interpolate(multi, size=tuple(ori_size), mode='bilinear', align_corners=True)
    return multi.permute(2, 3, 0, 1).reshape(1, ori_size[0], ori_size[1], obj_num, -1)


local_matching_proxy = local_matching  # AEM:1064-1156 is a verbatim copy of AEM:968-1060
