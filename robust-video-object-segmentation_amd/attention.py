"""Drop-in mirror of the hot-path parts of networks/layers/attention.py ("ATT") on the HIP library:
``IA_gate`` (ATT:7-17, the FiLM-style per-object modulation) and the masked mean pooling that
produces the k = 1 proxies and the 4C "IA head" (ATT:134-189).  Parameter name ``IA`` is part of
the checkpoint surface and is kept.
"""
import torch
from torch import nn

from . import ops


class IA_gate(nn.Module):
    """ATT:7-17.  forward(x [O,c,h,w], IA_head [O,D]) -> x * (1 + tanh(IA(IA_head)))[:, :, None, None]"""

    def __init__(self, in_dim, out_dim):
        super(IA_gate, self).__init__()
        self.IA = nn.Linear(in_dim, out_dim)

    def forward(self, x, IA_head):
        ops.inference_only("IA_gate", x, IA_head, self.IA.weight, self.IA.bias)
        return ops.film_scale(x, IA_head, self.IA.weight.detach(), self.IA.bias.detach())   # ATT:13-16 in one launch


def _pool_inputs(embeddings, labels):
    """[1 or O, C, h, w] embeddings (+ [O,1,h,w] labels) per frame -> channel-last [F, hw, C], [F, O, hw]."""
    embs, labs = [], []
    for e, l in zip(embeddings, labels):
        if e.dim() != 4 or l.dim() != 4:
            raise ValueError("expected [N,C,h,w] embeddings and [O,1,h,w] labels")
        if e.size(0) != 1 and e.stride(0) != 0:
            # per-object embeddings that genuinely differ are not produced anywhere in the reference
            # (aocnet.py:281,300 pass one map broadcast over objects)
            if not bool((e == e[:1]).all()):
                raise NotImplementedError("aoc_amd: per-object embeddings in attention pooling")
        c = e.size(1)
        embs.append(e[0].reshape(c, -1).t())
        labs.append(l.reshape(l.size(0), -1))
    return torch.stack(embs).contiguous(), torch.stack(labs).contiguous()


def calculate_attention_head_for_eval_p_m(ref_embeddings, ref_labels, prev_embedding, prev_label, epsilon=1e-5):
    """ATT:155-189 -> (total_head [O,4C], ref_head_pos, ref_head_neg, prev_head_pos, prev_head_neg)."""
    ops.inference_only("calculate_attention_head_for_eval_p_m", prev_embedding, *ref_embeddings)
    re, rl = _pool_inputs(ref_embeddings, ref_labels)
    ref_pos, ref_neg = ops.masked_mean_pool(re, rl, epsilon)
    pe, pl = _pool_inputs([prev_embedding], [prev_label])
    prev_pos, prev_neg = ops.masked_mean_pool(pe, pl, epsilon)
    total_head = torch.cat([ref_pos, ref_neg, prev_pos, prev_neg], dim=1)     # ATT:188
    return total_head, ref_pos, ref_neg, prev_pos, prev_neg


def calculate_attention_head_p_m(ref_embedding, ref_label, prev_embedding, prev_label, epsilon=1e-5):
    """ATT:134-153 (training twin, one reference frame)."""
    return calculate_attention_head_for_eval_p_m([ref_embedding], [ref_label], prev_embedding, prev_label, epsilon)


def calculate_attention_head_for_eval(ref_embeddings, ref_labels, prev_embedding, prev_label, epsilon=1e-5):
    """ATT:102-132: the head only."""
    return calculate_attention_head_for_eval_p_m(ref_embeddings, ref_labels, prev_embedding, prev_label, epsilon)[0]


def calculate_attention_head(ref_embedding, ref_label, prev_embedding, prev_label, epsilon=1e-5):
    """ATT:79-100."""
    return calculate_attention_head_p_m(ref_embedding, ref_label, prev_embedding, prev_label, epsilon)[0]
