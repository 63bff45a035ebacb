"""CPU restatement of the mask-calibration side of the hot path (TEST INFRASTRUCTURE ONLY).

Reference files:
  ATT = /root/reference/AOC-Net/complete_project/AOCNet/networks/layers/attention.py
  CL  = /root/reference/AOC-Net/conditioning_layer.py  (== CLB:6-48)
  CLB = /root/reference/AOC-Net/complete_project/AOCNet/networks/aoc/conditioning_layer.py

Functions take explicit weight tensors (no nn.Module state) so that tests can feed the same
weights to the HIP path.

PARITY STATUS
* ``ia_gate``, ``attention_head_for_eval_p_m``, ``conditioning_layer`` (4-D input): pinned by
  golden vectors generated from the reference (tests/golden/make_golden.py).
* ``conditioning_block``: CLB:66-86 cannot execute as shipped (``CL_1``/``mlp_layer`` without
  ``self.``; ``CL_2``/``CL_3`` receive 2-D tensors that Conv2d rejects; with [N,D,1,1] inputs
  ``k = int(beta*1*1) = 0`` and CLB:36 raises IndexError).  Since round 6 the reference's OWN forward
  is run unmodified with the four missing names injected as module globals
  (tests/golden/make_golden_r6.py: ``CL_1`` = the reference layer, ``mlp_layer`` = its MLP,
  ``CL_2``/``CL_3`` = the repair below) and its output pins CLB:68-69, CLB:72 and CLB:81-84
  (tests/golden/conditioning_block_injected_*.npz).  **Still unpinned** -- nothing in the reference
  can execute it: the repair of the vector branch, a 2-D input v treated as the H=W=1 limit of
  Eq.7 with the gate identically 1, i.e. ``CL(v) = mlp(v)``; ``CL_3`` is sized by ``proxy_dim``.
"""
import torch


# -------------------------------------------------------------------------- a10
def attention_head_for_eval_p_m(ref_embeddings, ref_labels, prev_embedding, prev_label, epsilon=1e-5):
    """ATT:155-189.  ref_embeddings: list of [1 or O, C, h, w]; ref_labels: list of [O,1,h,w];
    prev_embedding [O,C,h,w]; prev_label [O,1,h,w].  Returns (total_head [O,4C], ref_pos, ref_neg,
    prev_pos, prev_neg) each [O, C]."""
    tot_pos = tot_neg = tot_pos_n = tot_neg_n = 0.
    for emb, lab in zip(ref_embeddings, ref_labels):
        head = emb * lab                                                   # ATT:164
        pos = torch.sum(head, dim=(2, 3))
        neg = torch.sum(emb, dim=(2, 3)) - pos                             # ATT:166
        tot_pos = tot_pos + pos
        tot_neg = tot_neg + neg
        tot_pos_n = tot_pos_n + torch.sum(lab, dim=(2, 3))
        tot_neg_n = tot_neg_n + torch.sum(1. - lab, dim=(2, 3))
    ref_pos = tot_pos / (tot_pos_n + epsilon)                              # ATT:173-174
    ref_neg = tot_neg / (tot_neg_n + epsilon)
    head = prev_embedding * prev_label
    p_pos = torch.sum(head, dim=(2, 3))
    p_neg = torch.sum(prev_embedding, dim=(2, 3)) - p_pos
    p_pos = p_pos / (torch.sum(prev_label, dim=(2, 3)) + epsilon)
    p_neg = p_neg / (torch.sum(1. - prev_label, dim=(2, 3)) + epsilon)
    total = torch.cat([ref_pos, ref_neg, p_pos, p_neg], dim=1)            # ATT:188
    return total, ref_pos, ref_neg, p_pos, p_neg


# -------------------------------------------------------------------------- a11
def film_gain(head, weight, bias):
    """ATT:13-14:  a = 1 + tanh(Linear(head))      head [O,D], weight [c,D], bias [c] -> [O,c]"""
    return 1. + torch.tanh(torch.nn.functional.linear(head, weight, bias))


def ia_gate(x, head, weight, bias):
    """ATT:12-17 (IA_gate.forward): x * (1 + tanh(W head + b))[:, :, None, None]"""
    return film_gain(head, weight, bias).unsqueeze(-1).unsqueeze(-1) * x


# -------------------------------------------------------------------------- a12
def conditioning_gate_stats(z, phi_w, phi_b, beta_percentage=0.3):
    """CL:23-43.  z [N,C,H,W]; phi_w [C] (the 1x1 conv C->1), phi_b scalar.
    Returns (scores [N,HW], threshold [N], mask [N,HW] bool, gap [N,C])."""
    n, c, hgt, wid = z.shape
    zf = z.reshape(n, c, -1)
    s = (zf * phi_w.reshape(1, c, 1)).sum(1) + phi_b                      # CL:27  phi(z)
    k = int(beta_percentage * wid * hgt)                                   # CL:32
    top, _ = torch.topk(s, k=k, dim=-1, sorted=True)                      # CL:33
    thr = top[..., -1]
    mask = s > thr.unsqueeze(-1)                                           # CL:36  strict > : k-1 pixels
    gap = (zf * mask.unsqueeze(1)).mean(dim=-1)                            # CL:39-43 mean over ALL HW
    return s, thr, mask, gap


def conditioning_layer(z, phi_w, phi_b, mlp_w, mlp_b, beta_percentage=0.3):
    """CL:22-45 (paper Eq.7) with the missing ``self.`` on mlp_layer repaired.
    4-D z -> Linear(GAP(z * (phi(z) > kth_largest(phi(z)))));  2-D z (vector input, the
    conditioning_block repair) -> Linear(z)."""
    if z.dim() == 2:
        return torch.nn.functional.linear(z, mlp_w, mlp_b)
    _, _, _, gap = conditioning_gate_stats(z, phi_w, phi_b, beta_percentage)
    return torch.nn.functional.linear(gap, mlp_w, mlp_b)


# -------------------------------------------------------------------------- a13
def conditioning_block(x, proxy_ia_head, p, beta_percentage=0.3):
    """CLB:66-86 (paper Eq.5), repaired as described in the module docstring.
    ``p`` is a dict of weights: CL_1.{phi_w,phi_b,mlp_w,mlp_b}, CL_2.{mlp_w,mlp_b},
    CL_3.{mlp_w,mlp_b}, mlp_w [C, 2C+P], mlp_b [C]."""
    px1 = x.mean(dim=(2, 3))                                               # CLB:68 avg_pool2d over HxW
    x_delta = px1.sum(dim=0, keepdim=True) - px1                           # CLB:69
    c1 = conditioning_layer(x, p["CL_1.phi_w"], p["CL_1.phi_b"], p["CL_1.mlp_w"], p["CL_1.mlp_b"], beta_percentage)
    c2 = conditioning_layer(x_delta, None, None, p["CL_2.mlp_w"], p["CL_2.mlp_b"])
    c3 = conditioning_layer(proxy_ia_head, None, None, p["CL_3.mlp_w"], p["CL_3.mlp_b"])
    a = torch.nn.functional.linear(torch.cat([c1, c2, c3], dim=1), p["mlp_w"], p["mlp_b"])   # CLB:81
    a = 1. + torch.tanh(a)                                                 # CLB:82
    return a.unsqueeze(-1).unsqueeze(-1) * x                               # CLB:83-84


def gct_forward(x, alpha, gamma, beta, epsilon=1e-5, mode="l2", after_relu=False):
    """networks/layers/gct.py:17-36 restated.  Pinned by tests/golden/gct_*.npz, produced by the real ``GCT`` class
    (tests/golden/make_golden_r2.py registers empty stand-in modules for the ``networks.p2t`` import the file makes).
    x [N, C, H, W]; alpha/gamma/beta [1, C, 1, 1]."""
    if mode == "l2":
        embedding = (x.pow(2).sum((2, 3), keepdim=True) + epsilon).pow(0.5) * alpha
        norm = gamma / (embedding.pow(2).mean(dim=1, keepdim=True) + epsilon).pow(0.5)
    else:
        _x = x if after_relu else torch.abs(x)
        embedding = _x.sum((2, 3), keepdim=True) * alpha
        norm = gamma / (torch.abs(embedding).mean(dim=1, keepdim=True) + epsilon)
    gate = 1. + torch.tanh(embedding * norm + beta)
    return x * gate


def ia_logit(x, IA_head, weight, bias):
    """decoding_module.py:151-160 restated: IA_final = Linear(head_dim, C + 1) given by (weight, bias).
    Pinned by tests/golden/ia_logit.npz (the reference method itself, make_golden_r2.py)."""
    import torch.nn.functional as F
    n, c, h, w = x.size()
    xv = x.reshape(1, n * c, h, w)
    IA_output = F.linear(IA_head, weight, bias)
    IA_weight = IA_output[:, :c].reshape(n, c, 1, 1)
    IA_bias = IA_output[:, -1].reshape(-1)
    return F.conv2d(xv, weight=IA_weight, bias=IA_bias, groups=n).view(n, 1, h, w)


def augment_background_logit(fg_logit, bg_logit):
    """decoding_module.py:211-223: the background object's logit is raised by the minimum of the foreground objects'
    relative-background logits.  fg_logit, bg_logit [N, 1, H, W] -> [1, N, H, W]."""
    n = fg_logit.size(0)
    pred = fg_logit
    if n > 1:
        aug, _ = torch.min(bg_logit[1:n], dim=0, keepdim=True)
        pred = pred + torch.cat([aug, torch.zeros_like(aug).expand(n - 1, -1, -1, -1)], dim=0)
    return pred.permute(1, 0, 2, 3)


def dynamic_prehead(x, conv_w, conv_b, gn_w, gn_b, groups, eps=1e-5):
    """decoding_module.py:228-240: 1x1 conv -> GroupNorm(embed_dim / 4) -> ReLU.  Pinned by tests/golden/dynamic_prehead.npz."""
    import torch.nn.functional as F
    y = F.conv2d(x, conv_w.reshape(conv_w.shape[0], -1, 1, 1), conv_b)
    return F.relu(F.group_norm(y, int(groups), gn_w, gn_b, float(eps)))


def bottleneck(x, p, stride=1, dilation=1, groups=32, eps=1e-5, gct_eps=1e-5):
    """networks/layers/gct.py:38-90 (the decoder's residual block): GCT -> conv1x1 -> GN -> ReLU -> conv3x3 -> GN -> ReLU ->
    conv1x1 -> GN -> (+ residual, optionally conv1x1 + GN) -> ReLU.  ``p`` = the module's state_dict.  Pinned by
    tests/golden/bottleneck_64_128.npz.  Returns (out, stage1) with stage1 = ReLU(GN(conv1(GCT(x)))) (gct.py:69-72)."""
    import torch.nn.functional as F
    out = gct_forward(x, p["GCT1.alpha"], p["GCT1.gamma"], p["GCT1.beta"], gct_eps)
    out = F.relu(F.group_norm(F.conv2d(out, p["conv1.weight"]), groups, p["bn1.weight"], p["bn1.bias"], eps))
    stage1 = out
    out = F.conv2d(out, p["conv2.weight"], stride=stride, dilation=dilation, padding=dilation)
    out = F.relu(F.group_norm(out, groups, p["bn2.weight"], p["bn2.bias"], eps))
    out = F.group_norm(F.conv2d(out, p["conv3.weight"]), groups, p["bn3.weight"], p["bn3.bias"], eps)
    res = x
    if "downsample.0.weight" in p:
        res = F.group_norm(F.conv2d(x, p["downsample.0.weight"], stride=stride), groups, p["downsample.1.weight"], p["downsample.1.bias"], eps)
    return F.relu(out + res), stage1
