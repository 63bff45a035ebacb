#!/usr/bin/env python3
"""Round-4 golden vectors: the per-frame ORCHESTRATION and the eval-loop BOOKKEEPING, produced by RUNNING THE REFERENCE ITSELF
(build container only; see make_golden.py).

    python tests/golden/make_golden_r4.py

f-1  ``AOCNet.before_seghead_process`` (networks/aoc/aocnet.py:114-372) is called UNMODIFIED, as an unbound function, on a mock
     ``self`` (``cfg``, ``training=False``, ``epsilon``, ``bg_bias``, ``fg_bias``; ``dynamic_prehead`` = the reference's own
     ``DynamicPreHead`` behind a recording hook, ``dynamic_seghead`` = a recording hook).  aocnet.py:8 imports
     ``networks.p2t.decoding_module`` which the reference does not ship: that name is registered as an alias of the reference's own
     ``networks/aoc/decoding_module.py`` (the file that defines the two imported classes; make_golden_r2.load_decoder_modules imports
     it behind empty ``networks.p2t.center_module`` / ``conditioning_layer`` stand-ins).  Recorded per case: the inputs exactly as
     forward_for_eval hands them over (``[1, C, h, w]`` embeddings, FULL-RESOLUTION integer label maps incl. the label 125,
     ``gt_ids``), the biases, every ``kmeans2`` call, the 24-channel tensor handed to ``dynamic_prehead``, the prehead's output (the
     ``to_cat`` tensor of aocnet.py:362 is checked to be the concatenation it is, not stored), the attention head and the previous-frame mask handed to ``dynamic_seghead``.
f-2  ``Evaluator.evaluating`` (networks/engine/eval_manager_mm.py:160-394) is called UNMODIFIED on an ``Evaluator.__new__`` instance:
     the modules it imports that need cv2 / torchvision / datasets (``torchvision``, ``dataloaders.*``, ``networks.deeplab.deeplab``,
     ``utils.image``, ``utils.checkpoint``, ``utils.eval``) are empty stand-ins exposing the imported names, ``torch.Tensor.cuda`` is
     patched to the identity for the duration of the call, the dataset is a list of mock sequences shaped like ``VOS_Test`` samples
     after ``MultiToTensor`` and the model is a recording mock whose ``forward_for_eval`` logs what it is handed every frame (which
     embeddings are in the pool, every confident reference mask incl. 125, the previous mask) and returns scripted soft-max maps.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import syn, f16, km_arrays, KM_LOG, REF  # noqa: E402
import make_golden_r2 as mg2  # noqa: E402

CP = f"{REF}/complete_project/AOCNet"


def load_aocnet():
    """Import networks/aoc/aocnet.py unmodified (see the module docstring for the one aliased module name)."""
    gct, dm = mg2.load_decoder_modules()
    sys.modules.setdefault("networks.p2t.decoding_module", dm)
    aocnet = importlib.import_module("networks.aoc.aocnet")
    mtm = importlib.import_module("networks.layers.matching")
    mtm.kmeans2 = mg.recording_kmeans2                       # observe scipy's calls (forwards to the real function)
    return aocnet, dm


class Cfg:
    """The attributes before_seghead_process reads, with the values of configs/resnet101_aocnet.py:70-78,124-127."""
    MODEL_MULTI_LOCAL_DISTANCE = [2, 4, 6, 8, 10, 12]
    MODEL_LOCAL_DOWNSAMPLE = True
    MODEL_MATCHING_BACKGROUND = True
    MODEL_FLOAT16_MATCHING = False
    TEST_GLOBAL_CHUNKS = 4
    TEST_GLOBAL_ATROUS_RATE = 1
    TEST_LOCAL_ATROUS_RATE = 1
    TEST_LOCAL_PARALLEL = True


def save_raw(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"{name:34s} {os.path.getsize(path) / 1024:8.1f} KiB")


def upsample_labels(lab_hw, H, W, rng, noise=0):
    """A full-resolution integer label map whose nearest-neighbour down-sampling is NOT simply the stride-1 map: every full-resolution
    pixel takes the label of the map pixel under it, then `noise` random pixels get another label."""
    h, w = lab_hw.shape
    yy = (np.arange(H) * h // H).clip(0, h - 1)
    xx = (np.arange(W) * w // W).clip(0, w - 1)
    out = lab_hw[yy][:, xx].astype(np.int64)
    if noise:
        ys, xs = rng.randint(0, H, noise), rng.randint(0, W, noise)
        out[ys, xs] = rng.randint(0, int(lab_hw.max()) + 1, noise)
    return out


def run_frame(aocnet, dm, name, h, w, n_obj, ref_ids, prev_id, cur_id, seed, full=(97, 161), bg_bias=0.0, fg_bias=0.0, background=True,
              absent=None, uncertain=False, clip_seed=3):
    """One eval-mode call of before_seghead_process (batch 1, R = len(ref_ids) reference frames)."""
    C = 100
    cfgc = syn.ClipConfig("g", h, w, n_obj, 16, C, max(ref_ids + [prev_id, cur_id]) + 1)
    d = syn.make_clip(cfgc, clip_seed)
    emb, lab = f16(d["emb"]), d["lab"]
    rng = np.random.RandomState(seed)
    H, W = full
    ref_labels_full = []
    for j, i in enumerate(ref_ids):
        l = upsample_labels(lab[i], H, W, rng, noise=40)
        if absent is not None and j == absent[0]:
            l[l == absent[1]] = 0                                  # an object absent from one reference frame
        if uncertain and j == len(ref_ids) - 1:
            l[H // 3:H // 3 + 9, W // 4:W // 4 + 30] = 125          # the memory policy's "uncertain" label (eval_manager_mm.py:346)
        ref_labels_full.append(l)
    prev_label_full = upsample_labels(lab[prev_id], H, W, rng, noise=25)

    cfg = Cfg()
    cfg.MODEL_MATCHING_BACKGROUND = bool(background)
    torch.manual_seed(seed)
    in_dim = 24 if background else 17
    prehead = dm.DynamicPreHead(in_dim=in_dim, embed_dim=64)
    with torch.no_grad():
        for p in prehead.parameters():
            p.copy_(torch.from_numpy(f16(p.numpy())))
        prehead.bn.weight.copy_(torch.from_numpy(f16(np.random.RandomState(seed + 1).rand(64) + 0.5)))
        prehead.bn.bias.copy_(torch.from_numpy(f16(np.random.RandomState(seed + 2).randn(64) * 0.2)))
    cap = {}

    def prehead_hook(x):
        cap["pre_to_cat"] = x.detach().clone()
        y = prehead(x)
        cap["prehead_out"] = y.detach().clone()
        return y

    def seghead_hook(to_cat, attention_head, memory_prev, low_level_feat, to_cat_previous_frame):
        cap["to_cat"] = to_cat.detach().clone()
        cap["attention_head"] = attention_head.detach().clone()
        cap["seghead_prev_mask"] = to_cat_previous_frame.detach().clone()
        cap["low_level_shape"] = tuple(low_level_feat.shape)
        return torch.zeros(1, to_cat.shape[0], h, w), ["memory"]

    me = types.SimpleNamespace(cfg=cfg, training=False, epsilon=1e-5, dynamic_prehead=prehead_hook, dynamic_seghead=seghead_hook,
                               bg_bias=torch.full((1, 1, 1, 1), float(f16(bg_bias))), fg_bias=torch.full((1, 1, 1, 1), float(f16(fg_bias))))
    to_nchw = lambda e: torch.from_numpy(e).permute(2, 0, 1).unsqueeze(0).contiguous()
    ref_emb = [to_nchw(emb[i]) for i in ref_ids]
    ref_lab = [torch.from_numpy(l).view(1, 1, H, W) for l in ref_labels_full]
    prev_lab = torch.from_numpy(prev_label_full).view(1, 1, H, W)
    gt_ids = torch.tensor([n_obj - 1])
    low = torch.zeros(1, 8, h, w)
    KM_LOG.clear()
    np.random.seed(seed)
    with torch.no_grad():
        dic, boards, mem = aocnet.AOCNet.before_seghead_process(
            me, [[None, None]], ref_emb, to_nchw(emb[prev_id]), to_nchw(emb[cur_id]), ref_lab, prev_lab, gt_ids, current_low_level=low,
            tf_board=False)
    assert len(dic) == 1 and mem == [["memory"]]
    # aocnet.py:362: to_cat = cat(current embedding expanded over the objects, prehead output) -- checked here, not stored
    assert torch.equal(cap["to_cat"], torch.cat((to_nchw(emb[cur_id]).expand(n_obj, -1, -1, -1), cap["prehead_out"]), 1))
    km = km_arrays()
    for k in list(km):
        if k.endswith(("_labels", "_labels_it1", "_labels_it2")):
            km[k] = km[k].astype(np.int8)
    save_raw(name, in_ref=np.stack([emb[i] for i in ref_ids]).astype(np.float16), in_prev=emb[prev_id].astype(np.float16),
             in_cur=emb[cur_id].astype(np.float16), ref_labels_full=np.stack(ref_labels_full).astype(np.int16),
             prev_label_full=prev_label_full.astype(np.int16), n_obj=np.int32(n_obj), seed=np.int64(seed),
             bg_bias=np.float32(f16(bg_bias)), fg_bias=np.float32(f16(fg_bias)), background=np.int32(background),
             prehead_conv_w=prehead.conv.weight.detach().numpy(), prehead_conv_b=prehead.conv.bias.detach().numpy(),
             prehead_gn_w=prehead.bn.weight.detach().numpy(), prehead_gn_b=prehead.bn.bias.detach().numpy(),
             prehead_groups=np.int32(prehead.bn.num_groups), prehead_eps=np.float32(prehead.bn.eps),
             pre_to_cat=cap["pre_to_cat"].numpy(), prehead_out=cap["prehead_out"].numpy(),
             attention_head=cap["attention_head"].numpy(), seghead_prev_mask=cap["seghead_prev_mask"].numpy(), **km)


# ------------------------------------------------------------------------------------------------------------------ f-2
_EVALUATOR = []


def load_evaluator():
    """Import networks/engine/eval_manager_mm.py unmodified behind stand-ins for the modules that need cv2 / torchvision / datasets."""
    if _EVALUATOR:
        _EVALUATOR[0][1]["saved"].clear()
        return _EVALUATOR[0]
    if CP not in sys.path:
        sys.path.insert(0, CP)
    log = dict(saved=[])

    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms", Compose=lambda x: x)
    dl = stub("dataloaders")
    dl.datasets_m = stub("dataloaders.datasets_m", YOUTUBE_VOS_Test=object, DAVIS_Test=object)
    dl.custom_transforms = stub("dataloaders.custom_transforms")
    stub("networks.deeplab.deeplab", DeepLab=object)
    stub("utils.image", flip_tensor=lambda t, d: t.flip(d), save_mask=lambda m, p: log["saved"].append((p, m.clone())),
         save_matching_result=lambda *a, **k: None)
    stub("utils.checkpoint", load_network=lambda *a, **k: None)
    stub("utils.eval", zip_folder=lambda *a, **k: None)
    em = importlib.import_module("networks.engine.eval_manager_mm")
    _EVALUATOR.append((em, log))
    return em, log


class MockSequence(torch.utils.data.Dataset):
    """A sequence as VOS_Test hands it to the DataLoader after MultiRestrictSize + MultiToTensor (datasets_m.py:458-505,
    custom_transforms.py:465-486): a LIST with one sample per augmentation; 'current_label' [1, H, W] uint8 only on frames that
    carry ground truth."""

    def __init__(self, name, n_frames, H, W, gt, obj_nums):
        self.seq_name, self.n, self.H, self.W, self.gt, self.obj_nums = name, n_frames, H, W, gt, obj_nums

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        sample = {"current_img": torch.full((3, self.H, self.W), float(idx))}
        if idx in self.gt:
            sample["current_label"] = torch.from_numpy(self.gt[idx].astype(np.uint8))[None]
        sample["meta"] = {"seq_name": self.seq_name, "frame_num": self.n, "obj_num": int(self.obj_nums[idx]),
                          "current_name": f"{idx:05d}.jpg", "height": self.H, "width": self.W, "flip": False,
                          "obj_list": list(range(int(self.obj_nums[idx]) + 1))}
        return [sample]


class RecordingModel:
    """forward_for_eval's signature (aocnet.py:88); logs what the loop hands over, answers with the scripted soft-max maps."""

    def __init__(self, probs, C=4, hw=(5, 7)):
        self.probs, self.C, self.hw = probs, C, hw
        self.calls = []

    def eval(self):
        return self

    def forward_for_eval(self, memory_prev_list, ref_embeddings, ref_masks, prev_embedding, prev_mask, current_frame, pred_size, gt_ids):
        t = int(current_frame[0, 0, 0, 0].item())
        emb = torch.full((1, self.C, *self.hw), float(t))              # the frame index IS the embedding: the log shows who is in the pool
        self.calls.append(dict(
            frame=t, gt_ids=int(gt_ids[0]), pred_size=(int(pred_size[0]), int(pred_size[1])),
            ref_frames=[int(e[0, 0, 0, 0].item()) for e in ref_embeddings],
            ref_masks=[m.detach().clone().reshape(m.shape[-2], m.shape[-1]).to(torch.int64).numpy() for m in ref_masks],
            prev_frame=None if prev_embedding is None else int(prev_embedding[0, 0, 0, 0].item()),
            prev_mask=None if prev_mask is None else prev_mask.detach().clone().reshape(prev_mask.shape[-2], prev_mask.shape[-1]).to(torch.int64).numpy()))
        if prev_embedding is None:
            return None, emb, memory_prev_list
        return torch.from_numpy(self.probs[t])[None].clone(), emb, memory_prev_list


def run_eval_loop(name, n_frames, H, W, n_ch, gt, obj_nums, seed, mem_every=5, unc_ratio=1.0):
    em, log = load_evaluator()
    rng = np.random.RandomState(seed)
    probs = {}
    for t in range(1, n_frames):
        logits = rng.randn(n_ch, H, W).astype(np.float32) * 2.5
        logits[min(t % n_ch, n_ch - 1), H // 4:H // 2, W // 4:W // 2] += 4.0     # a confident region that moves between the channels
        probs[t] = torch.softmax(torch.from_numpy(logits), dim=0).numpy()
    model = RecordingModel(probs)
    ev = em.Evaluator.__new__(em.Evaluator)
    ev.cfg = types.SimpleNamespace(BLOCK_NUM=2, TEST_WORKERS=0)
    ev.mem_every, ev.unc_ratio, ev.gpu = mem_every, unc_ratio, 0
    ev.model = model
    ev.dataset = [MockSequence(name, n_frames, H, W, gt, obj_nums)]
    ev.result_root, ev.source_folder, ev.zip_dir = "/tmp/aoc_golden_r4", "/tmp/aoc_golden_r4", "/tmp/aoc_golden_r4.zip"
    real_cuda, real_empty = torch.Tensor.cuda, torch.cuda.empty_cache
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    import warnings
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ev.evaluating()
    finally:
        torch.Tensor.cuda, torch.cuda.empty_cache = real_cuda, real_empty
    out = dict(n_frames=np.int32(n_frames), n_ch=np.int32(n_ch), mem_every=np.int32(mem_every), unc_ratio=np.float32(unc_ratio),
               gt_frames=np.array(sorted(gt), np.int32), obj_nums=np.array(obj_nums, np.int32),
               probs=np.stack([probs[t] for t in range(1, n_frames)]))
    for t, g in gt.items():
        out[f"gt{t}"] = g.astype(np.int16)
    assert len(model.calls) == n_frames and [c["frame"] for c in model.calls] == list(range(n_frames))
    for c in model.calls:
        t = c["frame"]
        out[f"f{t}_ref_frames"] = np.array(c["ref_frames"], np.int32)
        out[f"f{t}_ref_masks"] = np.stack(c["ref_masks"]).astype(np.int16) if c["ref_masks"] else np.zeros((0, H, W), np.int16)
        out[f"f{t}_prev_frame"] = np.int32(-1 if c["prev_frame"] is None else c["prev_frame"])
        out[f"f{t}_prev_mask"] = (c["prev_mask"] if c["prev_mask"] is not None else np.zeros((0, W))).astype(np.int16)
    assert len(log["saved"]) == n_frames - 1
    out["saved_labels"] = np.stack([m.reshape(H, W).to(torch.int64).numpy() for _, m in log["saved"]]).astype(np.int16)
    save_raw(name, **out)


def main():
    aocnet, dm = load_aocnet()
    # ---------------------------------------------------------------- f-1: before_seghead_process, eval branch
    run_frame(aocnet, dm, "frame_R1_O2", 24, 40, 2, [0], 1, 2, seed=101, full=(95, 159))
    run_frame(aocnet, dm, "frame_R3_O4_bias", 24, 40, 4, [0, 2, 3], 4, 5, seed=102, full=(97, 161), bg_bias=0.25, fg_bias=-0.5)
    run_frame(aocnet, dm, "frame_R2_O3_absent_unc", 24, 40, 3, [0, 2], 3, 4, seed=103, full=(96, 157), bg_bias=-0.125, fg_bias=0.375,
              absent=(1, 2), uncertain=True)
    run_frame(aocnet, dm, "frame_R2_O3_nobg", 25, 37, 3, [0, 1], 2, 3, seed=104, full=(99, 146), fg_bias=0.5, background=False)

    # ---------------------------------------------------------------- f-2: Evaluator.evaluating
    H, W = 21, 29
    rng = np.random.RandomState(7)
    g0 = np.zeros((H, W), np.int64)
    g0[3:9, 4:12] = 1
    g0[11:18, 15:26] = 2
    # frame 7 carries ground truth that introduces object 3 (YouTube-VOS style, eval_manager_mm.py:287-290); channel 4 is never seen
    g7 = np.zeros((H, W), np.int64)
    g7[1:6, 18:27] = 3
    run_eval_loop("eval_loop_join_obj3", 14, H, W, 5, {0: g0, 7: g7}, [2] * 7 + [3] * 7, seed=201, mem_every=5, unc_ratio=1.0)
    # no new object; tighter uncertainty threshold; MEM_EVERY = 3
    run_eval_loop("eval_loop_mem3", 13, H, W, 3, {0: g0}, [2] * 13, seed=202, mem_every=3, unc_ratio=0.6)


if __name__ == "__main__":
    main()
