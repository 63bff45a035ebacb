"""Closed loop over several frames: matching -> 24-channel tensor -> DynamicPreHead -> (a fixed linear read-out standing in for the
conv decoder) -> class probabilities -> eval-loop memory policy -> reference pool of the next frame.  The GPU pipeline
(libaoc_hip.so through its Python mirrors) and the CPU oracle run the same loop on the same inputs, initial rows and read-out
weights; the masks of every frame must agree within the 1e-3 IoU budget of BASELINE.json's north_star, the k-means
assignments behind them being bit-identical as long as the pools are."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aoc():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import aoc_amd
    aoc_amd._lib.lib()
    return aoc_amd


def _iou(a, b, n_obj):
    vals = []
    for o in range(n_obj):
        pa, pb = a == o, b == o
        union = int((pa | pb).sum())
        vals.append(1.0 if union == 0 else int((pa & pb).sum()) / union)
    return float(np.mean(vals))


@pytest.mark.parametrize("cfg_name,T", [("tiny", 7), ("cfg2", 4)])
def test_closed_loop_masks_within_iou_budget(aoc, cfg_name, T):
    """tiny: 7 frames, three pool changes.  cfg2 (round 6): BASELINE.json configs[1]'s full size -- 121x213 maps, 3 objects + background, K = 16 -- for
    3 frames with a pool change in between (R = 1, 1, 2), every frame against the oracle's whole frame (about 20 s of CPU), and the per-frame k-means
    assignments of the pool compared BIT FOR BIT with the oracle's (== scipy's) while the two loops' pools agree."""
    from oracle import eval_loop as oe
    from oracle import hotpath as ohot
    from oracle import kmeans as okm
    syn, hot = aoc.synthetic, aoc.hotpath
    cfg = syn.CONFIGS[cfg_name]
    clip = syn.make_clip(cfg, 11, frames=T)
    n_obj, h, w, C = cfg.n_obj, cfg.h, cfg.w, cfg.c
    H, W = h * 4, w * 4
    emb = [torch.from_numpy(e) for e in clip["emb"]]
    gt0 = torch.from_numpy(np.kron(clip["lab"][0], np.ones((4, 4), np.int64))[:H, :W].astype(np.int32))
    bias = torch.zeros(n_obj)
    mc = hot.MatchingConfig()

    # read-out: pre-head (1x1 conv + GroupNorm + ReLU, the reference's DynamicPreHead) and one linear map per pixel to a logit
    torch.manual_seed(3)
    pre = hot.DynamicPreHead(in_dim=24, embed_dim=64)
    readout = torch.nn.Linear(64, 1)
    with torch.no_grad():
        readout.weight.mul_(6.0)

    def decode_cpu(feat):                                     # feat [O, 24, h, w] -> probabilities [O, H, W]
        y = torch.relu(pre.bn(pre.conv(feat)))                # decoding_module.py:236-240
        logit = readout(y.permute(0, 2, 3, 1)).squeeze(-1)    # [O, h, w]
        logit = torch.nn.functional.interpolate(logit[None], size=(H, W), mode="bilinear", align_corners=True)[0]
        return torch.softmax(logit, dim=0)

    pre_gpu = hot.DynamicPreHead(in_dim=24, embed_dim=64).cuda()
    pre_gpu.load_state_dict(pre.state_dict())
    readout_gpu = torch.nn.Linear(64, 1).cuda()
    readout_gpu.load_state_dict(readout.state_dict())

    def decode_gpu(feat):
        y = pre_gpu(feat)
        logit = readout_gpu(y.permute(0, 2, 3, 1)).squeeze(-1)
        logit = torch.nn.functional.interpolate(logit[None], size=(H, W), mode="bilinear", align_corners=True)[0]
        return torch.softmax(logit, dim=0)

    gpu = aoc.eval_loop.MemoryPolicy(mem_every=2, unc_ratio=0.9)
    cpu = oe.MemoryPolicy(mem_every=2, unc_ratio=0.9)
    gpu.start(emb[0].cuda(), gt0.cuda())
    cpu.start(emb[0], gt0.long())
    ious = []
    checked_pools = 0
    with torch.no_grad():
        for t in range(1, T):
            # the same initial rows on both sides, drawn from the CPU pool's row counts (as scipy would from numpy's RandomState)
            c_ref_lab = torch.stack([oe.label_onehot_nearest(m, h, w, n_obj) for m in cpu.ref_mask_confident])
            counts = [int(c_ref_lab[..., o].sum()) for o in range(n_obj)]
            rows = syn.kmeans_init_rows(1000 + t, counts, 16)
            g_ref_emb, g_ref_lab, g_prev_emb, g_prev_lab = gpu.reference_pool(h, w, n_obj)
            same_pool = torch.equal(g_ref_lab.cpu(), c_ref_lab)
            feat_g, _, aux = hot.proto_mask_features(mc, g_ref_emb, g_ref_lab, g_prev_emb, g_prev_lab, emb[t].cuda(), bias.cuda(), init_rows=rows)
            c_prev_lab = oe.label_onehot_nearest(cpu.prev_mask, h, w, n_obj)
            feat_c, _ = ohot.proto_mask_features(torch.stack(cpu.ref_embeddings), c_ref_lab, cpu.prev_embedding, c_prev_lab, emb[t], bias,
                                                 init_rows=rows)
            if same_pool:
                np.testing.assert_allclose(feat_g.cpu().numpy(), feat_c.numpy(), rtol=0, atol=5e-6)
                # bit-exact argmin cluster assignments (north_star) on this frame's pool, at this size, with this frame's initial rows
                cp = aoc.matching.cluster_proxies(g_ref_emb.reshape(-1, C), g_ref_lab.reshape(-1, n_obj), 16, rows)
                offs = cp["prep"].obj_offsets.cpu().numpy()
                got_lab, got_cen = cp["labels"].cpu().numpy(), cp["centroids"].cpu().numpy()
                pool_rows, flat_lab = torch.stack(cpu.ref_embeddings).reshape(-1, C).numpy(), c_ref_lab.reshape(-1, n_obj)
                ids = torch.where(flat_lab.sum(1) > 0, flat_lab.argmax(1), torch.full((flat_lab.shape[0],), -1)).numpy()   # uncertain pixels (125) belong to nobody
                kk = 16
                for o in range(n_obj):
                    kk = min(kk, counts[o])
                    if kk == 0:
                        continue
                    x = pool_rows[ids == o]
                    cb, l, _ = okm.kmeans2_matrix(x, x[rows[o]], 20)
                    assert np.array_equal(got_lab[offs[o]:offs[o + 1]], l), f"frame {t}, object {o}: k-means labels"
                    assert np.array_equal(got_cen[o][:kk], cb), f"frame {t}, object {o}: code book"
                checked_pools += 1
            lab_g, _, _ = gpu.update(emb[t].cuda(), decode_gpu(feat_g))
            lab_c, _, _ = cpu.update(emb[t], decode_cpu(feat_c))
            ious.append(_iou(lab_g.cpu().numpy(), lab_c.numpy(), n_obj))
    assert min(ious) >= 1.0 - 1e-3, ious
    assert checked_pools >= 1                                  # the loops' pools agreed at least on the first frame
    assert len(gpu.ref_embeddings) == len(cpu.ref_embeddings) == 1 + (T - 1) // 2
