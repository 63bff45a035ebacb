"""CPU restatement of the matching half of ``AOCNet.before_seghead_process`` (TEST INFRASTRUCTURE ONLY).

Reference: /root/reference/AOC-Net/complete_project/AOCNet/networks/aoc/aocnet.py:114-372 (eval branch, batch size 1).
PINNED by the reference itself: ``tests/golden/make_golden_r4.py`` imports aocnet.py unmodified (its one missing import,
``networks.p2t.decoding_module``, is aliased to the reference's own ``networks/aoc/decoding_module.py``), calls
``AOCNet.before_seghead_process`` on a mock ``self`` and records the 24-channel tensor handed to ``DynamicPreHead``, the prehead's
output and the attention head handed to the decoder (``tests/golden/frame_*.npz``; ``tests/test_oracle_golden.py::
test_frame_orchestration_vs_reference``).  Every callee is the oracle function that is pinned by its own golden vectors.
"""
import torch

from . import calibration as ocal
from . import matching as om


def proto_mask_features(ref_emb, ref_labels, prev_emb, prev_labels, cur_emb, dis_bias, multi_local_distance=(2, 4, 6, 8, 10, 12),
                        epsilon=1e-5, matching_background=True, init_rows=None, cluster_levels=None):
    """ref_emb [R,h,w,C], ref_labels [R,h,w,O] one-hot float, prev_emb/cur_emb [h,w,C], prev_labels [h,w,O].
    Returns (features [O, 24, h, w], attention_head [O, 4C]).  ``cluster_levels`` (e.g. (8, 16, 32), BASELINE.json
    configs[2]) runs the adaptive-proxy branch once per level and concatenates the 2-channel results in level order
    (22 + 2 * levels channels); ``init_rows`` is then a list per level."""
    R, h, w, C = ref_emb.shape
    O = ref_labels.shape[-1]
    refs = [ref_emb[i] for i in range(R)]
    labs = [ref_labels[i] for i in range(R)]
    bias = dis_bias.reshape(-1)
    g_fg = om.global_matching_for_eval(refs, cur_emb, labs, 4, bias)                               # aocnet.py:196
    if cluster_levels is None:
        g_cl = om.global_matching_for_eval_cluster(refs, cur_emb, labs, 4, bias, init_rows=init_rows)  # aocnet.py:242
    else:
        g_cl = om.global_matching_for_eval_cluster(refs, cur_emb, labs, 4, bias, init_rows=init_rows, cluster_num=list(cluster_levels))
    l_fg = om.local_matching(prev_emb, cur_emb, prev_labels, bias, list(multi_local_distance))     # aocnet.py:255
    ref_e = [e.permute(2, 0, 1).unsqueeze(0) for e in refs]
    ref_l = [l.permute(2, 0, 1).unsqueeze(1) for l in labs]
    prev_l = prev_labels.permute(2, 0, 1).unsqueeze(1)                                              # to_cat_previous_frame
    head, ref_pos, _, prev_pos, _ = ocal.attention_head_for_eval_p_m(                               # aocnet.py:297
        ref_e, ref_l, prev_emb.permute(2, 0, 1).unsqueeze(0).expand(O, -1, -1, -1), prev_l, epsilon)
    g_px = om.global_matching_for_eval_proxy(ref_pos, cur_emb, labs, 4, bias)                       # aocnet.py:314
    proxy_map = torch.matmul(prev_labels, prev_pos)                                                 # aocnet.py:325
    l_px = om.local_matching(proxy_map, cur_emb, prev_labels, bias, list(multi_local_distance))     # aocnet.py:328
    if g_cl.shape[-1] == 1:      # all-unlabelled early-out returns one channel (AEM:588-589); cannot be concatenated
        raise ValueError("reference pool has no labelled pixel")
    to_g_px = g_px.squeeze(0).permute(2, 3, 0, 1)                                                   # aocnet.py:341-345
    to_g_cl = g_cl.squeeze(0).permute(2, 3, 0, 1)
    to_g_fg = g_fg.squeeze(0).permute(2, 3, 0, 1)
    to_l_px = l_px.squeeze(0).permute(2, 3, 0, 1)
    to_l_fg = l_fg.squeeze(0).permute(2, 3, 0, 1)
    pre = torch.cat((to_g_fg, to_g_cl, to_g_px, to_l_fg, to_l_px, prev_l), 1)                       # aocnet.py:355
    if matching_background:
        g_bg = om.foreground2background(to_g_fg, O)                                                 # aocnet.py:350
        resh = to_l_fg.permute(0, 2, 3, 1).unsqueeze(1)
        l_bg = om.foreground2background(resh, O).permute(0, 4, 2, 3, 1).squeeze(-1)                 # aocnet.py:351-353
        pre = torch.cat([pre, l_bg, g_bg], 1)                                                       # aocnet.py:358
    return pre, head


def before_seghead_process_eval(ref_emb, ref_labels_full, prev_emb, prev_label_full, cur_emb, n_obj, bg_bias, fg_bias, prehead=None,
                                multi_local_distance=(2, 4, 6, 8, 10, 12), epsilon=1e-5, matching_background=True, init_rows=None):
    """aocnet.py:114-372 as forward_for_eval drives it (batch 1), from the FULL-RESOLUTION integer label maps.

    ref_emb [R,h,w,C]; ref_labels_full [R,H,W] int (125 = uncertain, matches no object); prev_emb / cur_emb [h,w,C];
    prev_label_full [H,W]; ``prehead`` = dict(conv_w, conv_b, gn_w, gn_b, groups, eps) or None.
    Returns (pre_to_cat [O,24,h,w], attention_head [O,4C], prehead output [O,E,h,w] or None)."""
    from .eval_loop import label_onehot_nearest
    R, h, w, C = ref_emb.shape
    ref_labels = torch.stack([label_onehot_nearest(ref_labels_full[r], h, w, n_obj) for r in range(R)])       # aocnet.py:127-131, 192
    prev_labels = label_onehot_nearest(prev_label_full, h, w, n_obj)                                          # aocnet.py:133-134, 151
    # aocnet.py:143-146: background bias for object 0, the foreground bias for every other object
    dis_bias = torch.cat([torch.as_tensor(bg_bias, dtype=torch.float32).reshape(1),
                          torch.as_tensor(fg_bias, dtype=torch.float32).reshape(1).expand(n_obj - 1)]) if n_obj > 1 else \
        torch.as_tensor(bg_bias, dtype=torch.float32).reshape(1)
    pre, head = proto_mask_features(ref_emb, ref_labels, prev_emb, prev_labels, cur_emb, dis_bias, multi_local_distance, epsilon,
                                    matching_background, init_rows)
    out = None
    if prehead is not None:
        out = ocal.dynamic_prehead(pre, prehead["conv_w"], prehead["conv_b"], prehead["gn_w"], prehead["gn_b"], prehead["groups"],
                                   prehead["eps"])                                                            # aocnet.py:360
    return pre, head, out
