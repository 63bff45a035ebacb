"""ORACLE (test infrastructure only): CPU restatement of the eval-loop memory policy of
networks/engine/eval_manager_mm.py:196-361 for the single-scale, no-flip configuration, with
networks/layers/shannon_entropy.py:10-13 and the label-map preparation of networks/aoc/aocnet.py:128-133,151.

PINNED by the reference itself: ``tests/golden/make_golden_r4.py`` imports eval_manager_mm.py unmodified (the modules it imports
that need cv2 / torchvision / the datasets are empty stand-ins, ``Tensor.cuda`` is the identity), runs ``Evaluator.evaluating`` on mock
sequences with a recording mock model and stores what the model is handed every frame -- which embeddings are in the pool, every
confident reference mask incl. the label 125, the previous mask -- and the label maps it saves (``tests/golden/eval_loop_*.npz``;
``tests/test_oracle_golden.py::test_eval_loop_bookkeeping_vs_reference``).  The entropy map has its own golden
(``shannon_entropy.npz``).  The arithmetic is torch CPU fp32 like the reference's (on its device).
"""
import numpy as np
import torch


def cal_shannon_entropy(preds):
    """shannon_entropy.py:10-13.  preds [B, n, H, W] -> uncertainty [B, 1, H, W]."""
    return -1.0 * torch.sum(preds * torch.log(preds + 1e-6), dim=1, keepdim=True)


def frame_decision(all_pred, label_all_list, join_label=None, unc_ratio=1.0):
    """all_pred [1, n_ch, H, W] probabilities.  Returns (pred_label [H, W] int64, pred_label_c [H, W] int64, uncertainty [H, W]).
    eval_manager_mm.py:253-265 (zero never-seen channels), :316-318 (argmax), :319-326 (join), :339-346 (uncertain -> 125)."""
    n_ch = all_pred.size(1)
    remake, exist = [], []
    for i in range(n_ch):
        if i not in label_all_list:
            remake.append(torch.zeros_like(all_pred[0, i]).unsqueeze(0))
        else:
            remake.append(all_pred[0, i].unsqueeze(0))
            exist.append(all_pred[0, i].unsqueeze(0))
    all_pred_z = torch.cat(remake, dim=0).unsqueeze(0)
    if exist:
        all_pred_exist = torch.cat(exist, dim=0).unsqueeze(0)
        uncertainty = cal_shannon_entropy(all_pred_exist)[0, 0]
    else:
        uncertainty = torch.zeros_like(all_pred[0, 0])
    pred_label = torch.argmax(torch.mean(all_pred_z, dim=0), dim=0)
    if join_label is not None:
        join_label = join_label.long()
        keep = (join_label == 0).long()
        pred_label = pred_label * keep + join_label * (1 - keep)
        join_uncertainty_map = (join_label < 0).long()
        uncertainty = uncertainty * keep + join_uncertainty_map * (1 - keep)
    uncertainty_region = (uncertainty > unc_ratio).long()
    pred_label_c = pred_label * (1 - uncertainty_region) + 125 * uncertainty_region
    return pred_label, pred_label_c, uncertainty


def label_onehot_nearest(label_hw, h, w, n_obj):
    """aocnet.py:128-133 (interpolate nearest, .int()) + :151 (== ref_obj_ids).float() -> [h, w, n_obj]."""
    lab = torch.nn.functional.interpolate(label_hw.float()[None, None], size=(h, w), mode="nearest").int()[0, 0]
    ids = torch.arange(0, n_obj).int().view(-1, 1, 1)
    return (lab[None] == ids).float().permute(1, 2, 0).contiguous()


class MemoryPolicy:
    """The list bookkeeping of eval_manager_mm.py:274-361 (one augmentation)."""

    def __init__(self, mem_every=5, unc_ratio=1.0):
        self.mem_every, self.unc_ratio = mem_every, unc_ratio
        self.ref_embeddings, self.ref_masks, self.ref_mask_confident = [], [], []
        self.prev_embedding = self.prev_mask = None
        self.label_all_list = []
        self.frame_idx = 0

    def _see(self, gt):
        for i in np.unique(gt.cpu().numpy()).tolist():
            if i not in self.label_all_list:
                self.label_all_list.append(i)

    def start(self, embedding, gt_label):
        self._see(gt_label)
        self.ref_embeddings.append(embedding)
        self.ref_masks.append(gt_label)
        self.ref_mask_confident.append(gt_label)
        self.prev_embedding, self.prev_mask = embedding, gt_label
        self.frame_idx = 1

    def update(self, embedding, probs, gt_label=None):
        label, conf, unc = frame_decision(probs[None], self.label_all_list, gt_label, self.unc_ratio)
        if gt_label is not None:
            self._see(gt_label)
            self.ref_embeddings.append(embedding)
            self.ref_masks.append(label)
            self.ref_mask_confident.append(conf)
        elif self.mem_every > -1 and self.frame_idx % self.mem_every == 0:
            self.ref_embeddings.append(embedding)
            self.ref_masks.append(label)
            self.ref_mask_confident.append(conf)
        self.prev_embedding, self.prev_mask = embedding, label
        self.frame_idx += 1
        return label, conf, unc
