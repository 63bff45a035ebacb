"""CPU restatement of the DAVIS-2017 evaluation measures J and F (TEST INFRASTRUCTURE ONLY).

The reference repository contains no J / F code: it writes palette PNGs and points to the official DAVIS toolkit / CodaLab servers
(/root/reference/README.md:110); its only in-repo IoU is utils/metric.py:3-34 (training logs).  The algorithm therefore lives in a
third-party package that is absent from /root/reference: davis2017-evaluation (davis2017/metrics.py: db_eval_iou, db_eval_boundary,
_seg2bmap), unpinned by the reference (no requirements file names it).  PARITY UNPINNED: this file restates the published definitions
and the GPU kernel (aoc_mask_jf_accumulate) is checked against it; there is no golden vector.

  J  (db_eval_iou)      |A & B| / |A | B|, 1 when both masks are empty.
  F  (db_eval_boundary) boundaries by _seg2bmap (a pixel is a boundary pixel when it differs from its east, south or south-east
                        neighbour; last row / column use the east / south neighbour only; the corner is never a boundary),
                        both dilated with skimage.morphology.disk(bound_pix), bound_pix = ceil(0.008 * hypot(H, W));
                        precision = |pred_b & dil(gt_b)| / |pred_b|, recall = |gt_b & dil(pred_b)| / |gt_b| with the toolkit's
                        conventions for empty boundaries; F = 2 P R / (P + R) (0 when P + R = 0).
"""
import numpy as np


def db_eval_iou(annotation, segmentation):
    a, s = annotation.astype(bool), segmentation.astype(bool)
    union = np.sum(a | s)
    return 1.0 if union == 0 else float(np.sum(a & s)) / float(union)


def seg2bmap(seg):
    seg = seg.astype(bool)
    e = np.zeros_like(seg)
    s = np.zeros_like(seg)
    se = np.zeros_like(seg)
    e[:, :-1] = seg[:, 1:]
    s[:-1, :] = seg[1:, :]
    se[:-1, :-1] = seg[1:, 1:]
    b = seg ^ e | seg ^ s | seg ^ se
    b[-1, :] = seg[-1, :] ^ e[-1, :]
    b[:, -1] = seg[:, -1] ^ s[:, -1]
    b[-1, -1] = 0
    return b


def _disk(r):
    y, x = np.mgrid[-r:r + 1, -r:r + 1]
    return (x * x + y * y) <= r * r


def _dilate(b, r):
    out = np.zeros_like(b)
    H, W = b.shape
    d = _disk(r)
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            if not d[dy + r, dx + r]:
                continue
            ys, ye = max(0, dy), min(H, H + dy)
            xs, xe = max(0, dx), min(W, W + dx)
            out[ys - dy:ye - dy, xs - dx:xe - dx] |= b[ys:ye, xs:xe]
    return out


def db_eval_boundary(foreground_mask, gt_mask, bound_th=0.008):
    bound_pix = bound_th if bound_th >= 1 else int(np.ceil(bound_th * np.linalg.norm(foreground_mask.shape)))
    fg_b, gt_b = seg2bmap(foreground_mask), seg2bmap(gt_mask)
    fg_dil, gt_dil = _dilate(fg_b, bound_pix), _dilate(gt_b, bound_pix)
    gt_match, fg_match = gt_b & fg_dil, fg_b & gt_dil
    n_fg, n_gt = int(fg_b.sum()), int(gt_b.sum())
    if n_fg == 0 and n_gt > 0:
        precision, recall = 1.0, 0.0
    elif n_fg > 0 and n_gt == 0:
        precision, recall = 0.0, 1.0
    elif n_fg == 0 and n_gt == 0:
        precision, recall = 1.0, 1.0
    else:
        precision, recall = fg_match.sum() / float(n_fg), gt_match.sum() / float(n_gt)
    return 0.0 if precision + recall == 0 else 2.0 * precision * recall / (precision + recall)


def jf_sums(pred, gt, n_obj):
    """Sum over the foreground objects 1 .. n_obj-1 of J and of F."""
    sj = sum(db_eval_iou(gt == o, pred == o) for o in range(1, n_obj))
    sf = sum(db_eval_boundary(pred == o, gt == o) for o in range(1, n_obj))
    return sj, sf
