"""Eval-loop memory policy: the immediate caller of the matching path.

Counterpart of the per-sequence state machine in ``networks/engine/eval_manager_mm.py:196-361`` for the single-scale,
no-flip configuration the reference evaluates with (configs/resnet101_aocnet.py: TEST_MULTISCALE=[1.], TEST_FLIP=False):

* frame 0 seeds the reference pool with the ground-truth mask (:274-282);
* every later frame: channels of labels that never appeared in a ground-truth map are zeroed (:253-265), the label map is
  the arg-max (:316-318), a ground-truth map that introduces new objects overrides it where it is non-zero (:319-326);
* the *confident* map that the matching path sees replaces pixels whose Shannon entropy exceeds ``unc_ratio`` by the label
  125, which matches no object (:305-306, :339-346, :357-361; shannon_entropy.py:10-13);
* the frame joins the pool when it carries ground truth (:296-297) or every ``mem_every`` frames (:309-312);
* the previous-frame embedding / mask always advance (:314, :350-353).

All per-pixel work runs in ``libaoc_hip.so`` (``aoc_confident_labels``, ``aoc_label_onehot_nearest``); this class only keeps
the lists.  ``reference_pool(h, w)`` hands the pool over in the layout ``hotpath.proto_mask_features`` takes.
"""
import torch

from . import ops

UNCERTAIN_LABEL = 125


class MemoryPolicy:
    def __init__(self, mem_every=5, unc_ratio=1.0, max_obj=None):
        self.mem_every = int(mem_every)
        self.unc_ratio = float(unc_ratio)
        self.max_obj = max_obj
        self.reset()

    def reset(self):
        """eval_manager_mm.py:376-382: all per-sequence state is dropped between sequences."""
        self.ref_embeddings, self.ref_masks, self.ref_mask_confident = [], [], []
        self.prev_embedding = self.prev_mask = None
        self.label_all = set()
        self.frame_idx = 0

    def _see(self, gt_label):
        # :267-272 np.unique of the ground truth (a host read-back in the reference as well; only on frames that carry GT)
        self.label_all.update(int(v) for v in torch.unique(gt_label).tolist())

    def exist_bits(self):
        bits = 0
        for l in self.label_all:
            if 0 <= l < 32:
                bits |= 1 << l
        return bits

    def start(self, embedding, gt_label):
        """Frame 0 (:274-282).  embedding [h, w, C]; gt_label int [H, W]."""
        assert self.frame_idx == 0
        gt_label = gt_label.to(torch.int32).contiguous()
        self._see(gt_label)
        self.ref_embeddings.append(embedding)
        self.ref_masks.append(gt_label)
        self.ref_mask_confident.append(gt_label)
        self.prev_embedding, self.prev_mask = embedding, gt_label
        self.frame_idx = 1

    def update(self, embedding, probs, gt_label=None):
        """One frame after the first.  probs [n_ch, H, W] class probabilities (soft-max of the decoder logits);
        gt_label int [H, W] when the frame carries ground truth (new objects, :288-289).
        Returns (label [H, W] int32, confident [H, W] int32, entropy [H, W] float32)."""
        assert self.frame_idx > 0, "call start() with the first frame"
        n_ch, H, W = probs.shape
        join = None
        if gt_label is not None:
            join = gt_label.to(torch.int32).contiguous()
        # the reference updates label_all_list from the current GT only AFTER zeroing the channels (:253-272)
        label, confident, entropy = ops.confident_labels(probs.reshape(n_ch, H * W), self.exist_bits(), join, self.unc_ratio)
        label, confident, entropy = label.view(H, W), confident.view(H, W), entropy.view(H, W)
        if gt_label is not None:
            self._see(join)
            self.ref_embeddings.append(embedding)                                  # :296-297
            self.ref_masks.append(label)                                           # :333
            self.ref_mask_confident.append(confident)                              # :339-348
        elif self.mem_every > -1 and self.frame_idx % self.mem_every == 0:
            self.ref_embeddings.append(embedding)                                  # :309-312
            self.ref_masks.append(label)
            self.ref_mask_confident.append(confident)                              # :357-361
        self.prev_embedding, self.prev_mask = embedding, label                     # :314, :350-353
        self.frame_idx += 1
        return label, confident, entropy

    def reference_pool(self, h, w, n_obj):
        """The pool as proto_mask_features takes it: (ref_emb [R, h, w, C], ref_labels [R, h, w, O] one-hot float of the
        CONFIDENT maps, prev_emb, prev_labels [h, w, O])  (aocnet.py:128-156)."""
        ref_emb = torch.stack(self.ref_embeddings, dim=0)
        ref_lab = torch.stack([ops.label_onehot_nearest(m, h, w, n_obj) for m in self.ref_mask_confident], dim=0)
        return ref_emb, ref_lab, self.prev_embedding, ops.label_onehot_nearest(self.prev_mask, h, w, n_obj)
