"""GPU parity tests of the round-2 rows: multi-level proxies (cfg3), atrous pool flattening, training twins, full-size cfg1
against the reference-generated golden, full-size cfg3 / cfg4 against the oracle, decoder-side modules against goldens produced
by the reference classes.  Tolerances as in test_gpu_parity.py: k-means labels / code books bit-exact, features atol 5e-6."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ATOL = 5e-6


@pytest.fixture(scope="module")
def aoc():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import aoc_amd
    aoc_amd._lib.lib()      # raises if the HIP library is missing: no silent fallback
    return aoc_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _refs(g):
    return [dev(e) for e in g["in_ref"]], [dev(l) for l in g["lab_onehot"]]


def _check_km(g, cp, n_obj, levels):
    """Every kmeans2 call the reference made (level-major, objects in order): rows drawn, labels and code book bit-identical."""
    multi = len(levels) > 1
    offs = cp["seg_offsets"].cpu().numpy()
    lab = cp["labels"].cpu().numpy()
    cen = cp["centroids"].cpu().numpy()
    call = 0
    for li in range(len(levels)):
        seg_k = cp["seg_k"][li] if multi else cp["seg_k"]
        rows = cp["init_rows"][li] if multi else cp["init_rows"]
        for i in range(n_obj):
            k = seg_k[i]
            if k == 0:
                continue
            s = li * n_obj + i
            assert k == int(g[f"km{call}_k"]) and offs[s + 1] - offs[s] == int(g[f"km{call}_n"])
            assert np.array_equal(rows[i], g[f"km{call}_rows"])
            assert np.array_equal(lab[offs[s]:offs[s + 1]], g[f"km{call}_labels"]), f"labels of level {levels[li]} object {i}"
            assert np.array_equal(cen[s, :k], g[f"km{call}_centroid"]), f"code book of level {levels[li]} object {i}"
            call += 1
    assert call == int(g["km_calls"])


# ------------------------------------------------------------------------------------------ cluster levels (cfg3)
@pytest.mark.parametrize("name", ["cluster_K8_R1_O3", "cluster_K32_R2_O4", "cluster_levels_8_16_32_R2_O4", "cluster_levels_small_obj"])
def test_cluster_levels_golden(aoc, golden, name):
    g = golden(name)
    refs, labs = _refs(g)
    levels = [int(v) for v in g["levels"]]
    cn = levels if len(levels) > 1 else levels[0]
    np.random.seed(int(g["seed"]))
    out = aoc.matching.global_matching_for_eval_cluster(refs, dev(g["in_query"]), labs, 4, dev(g["in_bias"]).view(-1, 1, 1, 1), None, 1, False, 0,
                                                        cluster_num=cn)
    assert tuple(out.shape) == g["out"].shape
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=0, atol=ATOL)
    c, o = g["in_ref"].shape[-1], g["lab_onehot"].shape[-1]
    np.random.seed(int(g["seed"]))
    cp = aoc.matching.cluster_proxies(dev(g["in_ref"].reshape(-1, c)), dev(g["lab_onehot"].reshape(-1, o)), cn)
    _check_km(g, cp, o, levels)


def test_cluster_levels_hotpath_vs_oracle(aoc):
    """The orchestrated path with CLUSTER_LEVELS = [8, 16, 32]: 28-channel tensor, sync-free chain == host-sync chain == oracle."""
    from aoc_amd import hotpath, synthetic as syn
    from oracle import hotpath as ohot
    cfg = syn.CONFIGS["tiny"]
    clip = syn.make_clip(cfg, seed=8, frames=5)
    O = cfg.n_obj
    emb = torch.from_numpy(clip["emb"])
    lab = torch.from_numpy(np.stack([syn.one_hot(l, O) for l in clip["lab"]]))
    bias = torch.tensor([0.2, -0.1, 0.05])
    levels = [8, 16, 32]
    mc = hotpath.MatchingConfig(CLUSTER_LEVELS=levels)
    np.random.seed(43)
    want, want_head = ohot.proto_mask_features(emb[[0, 2]], lab[[0, 2]], emb[3], lab[3], emb[4], bias, cluster_levels=levels)
    np.random.seed(43)
    got, head, aux = hotpath.proto_mask_features(mc, emb[[0, 2]].cuda(), lab[[0, 2]].cuda(), emb[3].cuda(), lab[3].cuda(), emb[4].cuda(), bias.cuda())
    assert tuple(got.shape) == (O, 28, cfg.h, cfg.w) == tuple(want.shape)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=ATOL)
    np.testing.assert_allclose(head.cpu().numpy(), want_head.numpy(), rtol=5e-6, atol=1e-7)
    # the same initial rows through the sync-free chain (device-side sticky K per level) and through a 2-frame batch
    init = np.zeros((len(levels) * O, 32), np.int32)
    for li in range(len(levels)):
        for o, r in enumerate(aux["cluster"]["init_rows"][li]):
            if r is not None:
                init[li * O + o, :len(r)] = r
    b, hb, _ = hotpath.proto_mask_features(mc, emb[[0, 2]].cuda(), lab[[0, 2]].cuda(), emb[3].cuda(), lab[3].cuda(), emb[4].cuda(), bias.cuda(),
                                           cluster_state=dict(init_rows=dev(init)))
    assert torch.equal(got, b) and torch.equal(head, hb)
    side = torch.cuda.Stream()
    pair = hotpath.launch_cluster_proxies_batch(mc, emb[[0, 2]].cuda(), lab[[0, 2]].cuda(), [dev(init), dev(init)], side)
    for a in pair:
        c, _, _ = hotpath.proto_mask_features(mc, emb[[0, 2]].cuda(), lab[[0, 2]].cuda(), emb[3].cuda(), lab[3].cuda(), emb[4].cuda(), bias.cuda(),
                                              cluster_ahead=a)
        assert torch.equal(got, c)


def test_cfg3_fullsize_levels_vs_oracle(aoc):
    """BASELINE.json configs[2] at FULL size: 145x261 map, O = 6, K in {8, 16, 32}, R = 2 reference frames.  k-means labels and
    code books of all 18 (level, object) calls bit-exact against the C oracle, cluster features within 5e-6 of the oracle."""
    from aoc_amd import synthetic as syn
    from oracle import kmeans as okm
    from oracle import matching as om
    cfg = syn.CONFIGS["cfg3"]
    assert (cfg.h, cfg.w, cfg.n_obj) == (145, 261, 6)
    clip = syn.make_clip(cfg, seed=3, frames=7)
    O, C = cfg.n_obj, cfg.c
    levels = [8, 16, 32]
    ref_ids, q_id = [0, 5], 6
    emb = torch.from_numpy(clip["emb"])
    lab = torch.from_numpy(np.stack([syn.one_hot(l, O) for l in clip["lab"]]))
    bias = torch.tensor([0.0, 0.1, -0.1, 0.2, -0.2, 0.05])
    counts = [int(sum((clip["lab"][i] == o).sum() for i in ref_ids)) for o in range(O)]
    assert min(counts) > 32
    rows = [syn.kmeans_init_rows(100 + k, counts, k) for k in levels]
    refs, labs = [emb[i] for i in ref_ids], [lab[i] for i in ref_ids]
    want, prox = om.global_matching_for_eval_cluster(refs, emb[q_id], labs, 4, bias, init_rows=rows, cluster_num=levels, return_proxies=True)
    got = aoc.matching.global_matching_for_eval_cluster([r.cuda() for r in refs], emb[q_id].cuda(), [l.cuda() for l in labs], 4, bias.cuda(),
                                                        None, 1, False, 0, init_rows=rows, cluster_num=levels)
    assert tuple(got.shape) == (1, cfg.h, cfg.w, O, 6) == tuple(want.shape)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=ATOL)
    pool = torch.cat([r.reshape(-1, C) for r in refs]).cuda()
    lflat = torch.cat([l.reshape(-1, O) for l in labs]).cuda()
    cp = aoc.matching.cluster_proxies(pool, lflat, levels, rows)
    offs, labd, cen = cp["seg_offsets"].cpu().numpy(), cp["labels"].cpu().numpy(), cp["centroids"].cpu().numpy()
    for li, k in enumerate(levels):
        for o in range(O):
            s = li * O + o
            p = prox[li][o]
            assert np.array_equal(labd[offs[s]:offs[s + 1]], p["labels"]), f"K={k} object {o}: labels differ"
            assert np.array_equal(cen[s, :k], p["centroid"].numpy()), f"K={k} object {o}: code book differs"


def test_cfg3_fullsize_hotpath_frame(aoc):
    """One whole cfg3 frame through the orchestrator (28 channels) against the oracle on the cluster / proxy / local channels and on a
    sub-sample of the dense channel."""
    from aoc_amd import hotpath, synthetic as syn
    from oracle import matching as om
    cfg = syn.CONFIGS["cfg3"]
    clip = syn.make_clip(cfg, seed=4, frames=3)
    O, C, hw = cfg.n_obj, cfg.c, cfg.h * cfg.w
    levels = [8, 16, 32]
    mc = hotpath.MatchingConfig(CLUSTER_LEVELS=levels)
    emb = torch.from_numpy(clip["emb"])
    lab = torch.from_numpy(np.stack([syn.one_hot(l, O) for l in clip["lab"]]))
    bias = torch.zeros(O)
    counts = [int((clip["lab"][0] == o).sum()) for o in range(O)]
    rows = [syn.kmeans_init_rows(200 + k, counts, k) for k in levels]
    feat, head, _ = hotpath.proto_mask_features(mc, emb[:1].cuda(), lab[:1].cuda(), emb[1].cuda(), lab[1].cuda(), emb[2].cuda(), bias.cuda(), init_rows=rows)
    ch = hotpath.channel_slices(mc)
    assert tuple(feat.shape) == (O, 28, cfg.h, cfg.w)
    feat = feat.cpu()
    want = om.global_matching_for_eval_cluster([emb[0]], emb[2], [lab[0]], 4, bias, init_rows=rows, cluster_num=levels)
    np.testing.assert_allclose(feat[:, ch["cluster"]:ch["cluster"] + 6].numpy(), want[0].permute(2, 3, 0, 1).numpy(), rtol=0, atol=ATOL)
    want = om.local_matching(emb[1], emb[2], lab[1], bias, [2, 4, 6, 8, 10, 12])
    np.testing.assert_allclose(feat[:, ch["local"]:ch["local"] + 6].numpy(), want[0].permute(2, 3, 0, 1).numpy(), rtol=0, atol=ATOL)
    q = emb[2].reshape(-1, C)[::23]
    dn = om.proto_transform(om.nearest_neighbor_features_per_object(emb[0].reshape(-1, C), q, lab[0].reshape(-1, O)).squeeze(-1), bias.view(1, -1))
    np.testing.assert_allclose(feat[:, 0].reshape(O, hw)[:, ::23].t().numpy(), dn.numpy(), rtol=0, atol=ATOL)
    np.testing.assert_array_equal(feat[:, ch["prev_mask"]].numpy(), lab[1].permute(2, 0, 1).numpy())


def test_cfg4_three_conditioning_blocks_fullsize(aoc):
    """BASELINE.json configs[3]: 3 calibration iterations (conditioning blocks) on x [9, 256, 181, 321] against the oracle."""
    from oracle import calibration as ocal
    torch.manual_seed(4)
    N, C, H, W, P = 9, 256, 181, 321, 400
    blocks = [aoc.conditioning_layer.conditioning_block(C, P, 0.3).cuda() for _ in range(3)]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, C, H, W, generator=g)
    head = torch.randn(N, P, generator=g)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    y_gpu, y_cpu = x.cuda(), x
    with torch.no_grad():
        for blk in blocks:
            sd = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
            w = {"CL_1.phi_w": sd["CL_1.phi_layer.weight"].reshape(-1), "CL_1.phi_b": sd["CL_1.phi_layer.bias"],
                 "CL_1.mlp_w": sd["CL_1.mlp_layer.weight"], "CL_1.mlp_b": sd["CL_1.mlp_layer.bias"],
                 "CL_2.mlp_w": sd["CL_2.mlp_layer.weight"], "CL_2.mlp_b": sd["CL_2.mlp_layer.bias"],
                 "CL_3.mlp_w": sd["CL_3.mlp_layer.weight"], "CL_3.mlp_b": sd["CL_3.mlp_layer.bias"],
                 "mlp_w": sd["mlp_layer.weight"], "mlp_b": sd["mlp_layer.bias"]}
            # per block from the SAME input (a near-tie at the k-th largest score may legitimately select another pixel set after a
            # 1e-6 perturbation; chaining the GPU output into the oracle would test that sensitivity, not the kernels)
            want = ocal.conditioning_block(y_cpu, head, w, 0.3)
            got = blk(y_gpu, head.cuda())
            np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=2e-5, atol=2e-5)
            k = int(0.3 * H * W)
            gap, scores, thr = aoc.ops.cond_gate_pool(y_gpu, blk.CL_1.phi_layer.weight.detach().reshape(-1), blk.CL_1.phi_layer.bias.detach(), k, want_debug=True)
            s = scores.cpu().numpy()
            assert np.array_equal(thr.cpu().numpy(), np.sort(s, axis=1)[:, ::-1][:, k - 1]), "k-th largest is not the exact order statistic"
            y_cpu = want
            y_gpu = want.cuda()


# ------------------------------------------------------------------------------------------ atrous + training twins
@pytest.mark.parametrize("name,fn", [("cluster_atrous2", "cluster"), ("cluster_atrous2_objpix", "cluster"), ("dense_atrous2", "dense"),
                                     ("dense_atrous2_objpix", "dense"), ("dense_atrous3_even", "dense")])
def test_atrous_pool_flattening_golden(aoc, golden, name, fn):
    g = golden(name)
    refs, labs = _refs(g)
    rate, objpix = int(g["atrous_rate"]), int(g["atrous_obj_pixel_num"])
    np.random.seed(int(g["seed"]))
    f = aoc.matching.global_matching_for_eval_cluster if fn == "cluster" else aoc.matching.global_matching_for_eval
    keep = [l.clone() for l in labs]
    out = f(refs, dev(g["in_query"]), labs, 4, dev(g["in_bias"]).view(-1, 1, 1, 1), None, rate, False, objpix)
    assert tuple(out.shape) == g["out"].shape
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=0, atol=ATOL)
    assert all(torch.equal(a, b) for a, b in zip(keep, labs)), "the caller's label maps must not be written to"


@pytest.mark.parametrize("name", ["dense_train_twin_atrous2", "cluster_train_twin", "cluster_train_twin_atrous2", "cluster_train_twin_unlabelled"])
def test_training_twins_golden(aoc, golden, name):
    g = golden(name)
    rate, objpix = int(g["atrous_rate"]), int(g["atrous_obj_pixel_num"])
    np.random.seed(int(g["seed"]))
    args = (dev(g["in_ref"][0]), dev(g["in_query"]), dev(g["lab_onehot"][0]), 3, dev(g["in_bias"]).view(-1, 1, 1, 1), None, rate, False, objpix)
    out = aoc.matching.global_matching(*args) if name.startswith("dense") else aoc.matching.global_matching_cluster2(*args)
    assert tuple(out.shape) == g["out"].shape
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=0, atol=ATOL)


# ------------------------------------------------------------------------------------------ full-size cfg1 golden
def test_fullsize_cfg1_golden(aoc):
    """One full-size cfg1 frame recorded from the reference: k-means labels / code books bit-exact, every pixel of the cluster,
    dense and local outputs against the float16-compressed record, an exact float32 sub-sample at 5e-6, and the checksums."""
    from test_oracle_golden import check_fullsize, fullsize_inputs
    from aoc_amd import synthetic as syn
    with np.load(os.path.join(os.path.dirname(__file__), "golden", "fullsize_cfg1.npz")) as z:
        g = {k: z[k] for k in z.files}
    cfg, d = fullsize_inputs(g)
    O = cfg.n_obj
    e0, e1 = dev(d["emb"][0]), dev(d["emb"][1])
    l0 = dev(syn.one_hot(d["lab"][0], O))
    b = torch.zeros(O, 1, 1, 1).cuda()
    np.random.seed(int(g["seed"]))
    out = aoc.matching.global_matching_for_eval_cluster([e0], e1, [l0], 4, b, None, 1, False, 0)
    check_fullsize(g, "cluster", out.cpu().numpy()[0], ATOL)
    np.random.seed(int(g["seed"]))
    cp = aoc.matching.cluster_proxies(e0.reshape(-1, cfg.c), l0.reshape(-1, O))
    g2 = dict(g)
    for i in range(int(g["km_calls"])):
        g2[f"km{i}_labels"] = g[f"km{i}_labels"].astype(np.int32)
    _check_km(g2, cp, O, [16])
    out = aoc.matching.global_matching_for_eval([e0], e1, [l0], 16, b, None, 1, False, 0)
    check_fullsize(g, "dense", out.cpu().numpy()[0], ATOL)
    out = aoc.matching.local_matching(e0, e1, l0, b, [2, 4, 6, 8, 10, 12], None, 1, False, True, True)
    check_fullsize(g, "local", out.cpu().numpy()[0], ATOL)


# ------------------------------------------------------------------------------------------ decoder-side modules vs the reference classes
@pytest.mark.parametrize("name,mode,relu", [("gct_l2", "l2", False), ("gct_l1", "l1", False), ("gct_l1_relu", "l1", True)])
def test_gct_golden(aoc, golden, name, mode, relu):
    g = golden(name)
    m = aoc.gct.GCT(12, float(g["eps"]), mode, relu).cuda()
    with torch.no_grad():
        m.alpha.copy_(dev(g["in_alpha"]).view(1, -1, 1, 1))
        m.gamma.copy_(dev(g["in_gamma"]).view(1, -1, 1, 1))
        m.beta.copy_(dev(g["in_beta"]).view(1, -1, 1, 1))
        y = m(dev(g["in_x"]))
    np.testing.assert_allclose(y.cpu().numpy(), g["out"], rtol=2e-6, atol=1e-6)


def test_prehead_and_ia_logit_golden(aoc, golden):
    from aoc_amd import hotpath
    g = golden("dynamic_prehead")
    ph = hotpath.DynamicPreHead(24, 64).cuda()
    with torch.no_grad():
        ph.conv.weight.copy_(dev(g["in_conv_w"]))
        ph.conv.bias.copy_(dev(g["in_conv_b"]))
        ph.bn.weight.copy_(dev(g["in_gn_w"]))
        ph.bn.bias.copy_(dev(g["in_gn_b"]))
        assert ph.bn.num_groups == int(g["groups"])
        y = ph(dev(g["in_x"]))
    np.testing.assert_allclose(y.cpu().numpy(), g["out"], rtol=1e-5, atol=2e-6)
    g = golden("ia_logit")
    fin = torch.nn.Linear(40, 17).cuda()
    with torch.no_grad():
        fin.weight.copy_(dev(g["in_w"]))
        fin.bias.copy_(dev(g["in_b"]))
        y = aoc.gct.IA_logit(dev(g["in_x"]), dev(g["in_head"]), fin)
    np.testing.assert_allclose(y.cpu().numpy(), g["out"], rtol=1e-5, atol=2e-6)


# ------------------------------------------------------------------------------------------ device J / F metric and the sharded runner
def test_mask_jf_device_vs_oracle(aoc):
    """aoc_mask_jf_accumulate against the restated DAVIS measures (oracle/metrics.py), accumulated over several frames."""
    from aoc_amd import synthetic as syn
    from oracle import metrics as om
    rng = np.random.RandomState(2)
    H, W, O = 97, 141, 5
    jf = aoc.ops.MaskJF(torch.device("cuda"))
    want_j = want_f = 0.0
    cfg = syn.ClipConfig("m", H, W, O, 16, 4, 6)
    tracks = syn.blob_tracks(rng, H, W, O)
    for t in range(4):
        gt = syn.label_map(tracks, t, H, W)
        pred = syn.label_map(tracks, t + (2 if t else 0), H, W)            # frame 0: identical maps; later: shifted blobs
        if t == 3:
            pred[pred == 2] = 0                                            # an object the prediction misses entirely
            gt[gt == 4] = 0                                                # ... and one absent from the ground truth
        jf.add(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda(), O)
        sj, sf = om.jf_sums(pred, gt, O)
        want_j += sj
        want_f += sf
    tot = jf.totals()
    assert tot["frames"] == 4 and tot["objects"] == 4 * (O - 1)
    assert abs(tot["sum_j"] - want_j) < 1e-9 and abs(tot["sum_f"] - want_f) < 1e-9


def test_eval_runner_on_the_gpu(aoc):
    """The sharded runner (one rank) through the real hot path: a DAVIS-like and two YouTube-VOS-like (multi-level) sequences."""
    from aoc_amd import eval_runner as er
    specs = er.make_sequence_set("cfg5", scale=0.004, seed=1)              # 1 + 2 sequences
    assert [s.levels for s in specs] == [(16,), (8, 16, 32), (8, 16, 32)]
    np.random.seed(0)
    tot = er.eval_sharded(specs, 0, 1, torch.device("cuda"), max_frames=4)
    assert tot["frames"] == 9 and tot["ranks"] == 1 and tot["iou_count"] == sum(3 * (s.n_obj - 1) for s in specs)
    assert 0.0 < tot["mean_j"] <= 1.0 and 0.0 <= tot["mean_f"] <= 1.0
    # the same run again gives the same numbers (seeded clips, seeded read-out, numpy's RandomState for the k-means rows)
    np.random.seed(0)
    again = er.eval_sharded(specs, 0, 1, torch.device("cuda"), max_frames=4)
    assert again["sum_iou"] == tot["sum_iou"] and again["sum_f"] == tot["sum_f"]
    # three sequences in flight (one HIP stream, backend and metric accumulator each): the same per-sequence results, summed per lane
    np.random.seed(0)
    lanes = er.eval_sharded(specs, 0, 1, torch.device("cuda"), max_frames=4, lanes=3)
    assert lanes["frames"] == tot["frames"] and lanes["iou_count"] == tot["iou_count"] and lanes["objects"] == tot["objects"]
    assert abs(lanes["sum_iou"] - tot["sum_iou"]) < 1e-9 and abs(lanes["sum_f"] - tot["sum_f"]) < 1e-9


# ------------------------------------------------------------------------------------------ local atrous, use_float16=True
@pytest.mark.parametrize("name", ["local_atrous2_down_O3", "local_atrous3_nodown_O3"])
def test_local_atrous_golden(aoc, golden, name):
    g = golden(name)
    out = aoc.matching.local_matching(dev(g["in_prev"]), dev(g["in_query"]), dev(g["lab_onehot"]), dev(g["in_bias"]).view(-1, 1, 1, 1),
                                      [int(v) for v in g["mld"]], None, int(g["atrous_rate"]), False, bool(g["down"]), True)
    assert tuple(out.shape) == g["out"].shape
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=0, atol=ATOL)


def _f16_close(got, want, frac_exact=0.999, atol=4e-3):
    """use_float16 mode: the reference's distances are float16 TENSORS, so every step that rounds to float16 must round the same fp32 value.
    Since round 3 the device reproduces torch-CPU's arithmetic where it matters -- the float16 bilinear resize term by term (its fp32 values
    are float16 ties surprisingly often), dot products k-sequentially in fp32, squares rounded to float16 before they are summed -- and the
    dense and local outputs equal the reference's own `.half()` code to 1.2e-7 on every element.  What is left is the summation order of a
    dot product in the k = 1 proxy path (1 output of 2 880 off by one float16 ulp of a distance: 9e-4).  Stated tolerance: >= 99.9 % of the
    outputs equal to 2e-6, all within 4e-3."""
    diff = np.abs(got - want)
    assert diff.max() <= atol, diff.max()
    assert np.mean(diff <= 2e-6) >= frac_exact, np.mean(diff <= 2e-6)


def test_float16_mode_vs_reference_half_path(aoc, golden):
    """use_float16=True (the default argument of the reference functions): dense, local (plain and atrous) against the outputs of the
    reference's own `.half()` code on torch-CPU; the k = 1 proxy path (reference raises UnboundLocalError) against the oracle."""
    from oracle import matching as om
    g = golden("dense_fp16_R2_O3")
    refs, labs = _refs(g)
    out = aoc.matching.global_matching_for_eval(refs, dev(g["in_query"]), labs, 4, dev(g["in_bias"]).view(-1, 1, 1, 1))     # use_float16 defaults to True
    assert out.dtype == torch.float32 and tuple(out.shape) == g["out"].shape
    _f16_close(out.cpu().numpy(), g["out"])
    for name in ("local_fp16_down_O3", "local_atrous2_fp16_down_O3"):
        g = golden(name)
        out = aoc.matching.local_matching(dev(g["in_prev"]), dev(g["in_query"]), dev(g["lab_onehot"]), dev(g["in_bias"]).view(-1, 1, 1, 1),
                                          [int(v) for v in g["mld"]], None, int(g.get("atrous_rate", 1)), True, bool(g["down"]), True)
        _f16_close(out.cpu().numpy(), g["out"])
    g = golden("proxy_eval_O3")
    lab = g["lab_onehot"][0]
    want = om.global_matching_for_eval_proxy(torch.from_numpy(g["in_proxies"]), torch.from_numpy(g["in_query"]), [torch.from_numpy(lab.copy())], 4,
                                             torch.from_numpy(g["in_bias"]), None, 1, True, 0)
    out = aoc.matching.global_matching_for_eval_proxy(dev(g["in_proxies"]), dev(g["in_query"]), [dev(lab)], 4, dev(g["in_bias"]).view(-1, 1, 1, 1))
    _f16_close(out.cpu().numpy(), want.numpy())
    g = golden("cluster_fp16_R2_O3")
    refs, labs = _refs(g)
    out = aoc.matching.global_matching_for_eval_cluster(refs, dev(g["in_query"]), labs, 4, dev(g["in_bias"]).view(-1, 1, 1, 1))
    assert np.array_equal(out.cpu().numpy(), g["out"])                    # exactly 1.0 everywhere, two channels


def test_dense_more_than_16_objects(aoc):
    """O = 21 objects (general float labels): the split kernel covers <= 16, the exact kernel walks the objects 16 at a time."""
    from oracle import matching as om
    rng = np.random.RandomState(9)
    h, w, c, o = 19, 27, 100, 21
    ref = (np.maximum(rng.randn(2, h, w, c), 0) * 0.3).astype(np.float32)
    q = (np.maximum(rng.randn(h, w, c), 0) * 0.3).astype(np.float32)
    ids = rng.randint(0, o, size=(2, h, w))
    lab = (ids[..., None] == np.arange(o)).astype(np.float32)
    bias = (rng.rand(o).astype(np.float32) - 0.5)
    want = om.global_matching_for_eval([torch.from_numpy(r) for r in ref], torch.from_numpy(q), [torch.from_numpy(l) for l in lab], 4, torch.from_numpy(bias))
    got = aoc.matching.global_matching_for_eval([dev(r) for r in ref], dev(q), [dev(l) for l in lab], 4, dev(bias), None, 1, False, 0)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=ATOL)


def test_incremental_proxies_accuracy(aoc):
    """NON-PARITY mode as a library option: hotpath.IncrementalProxyBank clusters every reference frame once, when it joins
    the pool, and matches against the union of the per-frame code books.  Every channel but the two cluster channels is identical to the
    reference mode; the cluster channels disagree with a fresh whole-pool clustering no more than two fresh clusterings with different
    initial rows disagree with each other (object decision from the nearest-proxy channel; mean absolute feature difference within 2x)."""
    syn, hot = aoc.synthetic, aoc.hotpath
    cfg = syn.CONFIGS["cfg1"]
    clip = syn.make_clip(cfg, 9, frames=8)
    O = cfg.n_obj
    emb = torch.from_numpy(clip["emb"]).cuda()
    lab = torch.from_numpy(np.stack([syn.one_hot(l, O) for l in clip["lab"]])).cuda()
    mc = hot.MatchingConfig()
    bias = torch.zeros(O, device="cuda")
    side = torch.cuda.Stream()
    ch = hot.channel_slices(mc)
    pool_ids = [0, 2, 4]
    ref_emb, ref_lab = emb[pool_ids].contiguous(), lab[pool_ids].contiguous()

    def init_for(seed, ids):
        counts = [int(sum((clip["lab"][i] == o).sum() for i in ids)) for o in range(O)]
        rows = syn.kmeans_init_rows(seed, counts, 16)
        init = np.zeros((O, 16), np.int32)
        for o, r in enumerate(rows):
            if r is not None:
                init[o, :len(r)] = r
        return torch.from_numpy(init).cuda()

    bank = hot.IncrementalProxyBank(mc, O, cfg.c, capacity_frames=4, device=emb.device)
    for n, i in enumerate(pool_ids):
        bank.append(emb[i], lab[i], init_for(100 + n, [i]), side)
    assert bank.R == 3
    c0 = ch["cluster"]
    dec = lambda f: f[:, c0].argmin(0)
    agree_inc, agree_fresh, diff_inc, diff_fresh = [], [], [], []
    for t in (5, 6, 7):
        feats = []
        for seed in (10 + t, 20 + t):
            a = hot.launch_cluster_proxies(mc, ref_emb, ref_lab, init_for(seed, pool_ids), side)
            f, _, _ = hot.proto_mask_features(mc, ref_emb, ref_lab, emb[t - 1], lab[t - 1], emb[t], bias, cluster_ahead=a)
            feats.append(f)
        f_i, _, _ = hot.proto_mask_features(mc, ref_emb, ref_lab, emb[t - 1], lab[t - 1], emb[t], bias, cluster_ahead=bank.handle(ref_lab))
        torch.cuda.synchronize()
        f_a, f_b = feats
        other = [i for i in range(f_a.shape[1]) if i not in (c0, c0 + 1)]
        assert torch.equal(f_a[:, other], f_i[:, other])
        agree_fresh.append(float((dec(f_a) == dec(f_b)).float().mean()))
        agree_inc.append(float((dec(f_a) == dec(f_i)).float().mean()))
        diff_fresh.append(float((f_a[:, c0:c0 + 2] - f_b[:, c0:c0 + 2]).abs().mean()))
        diff_inc.append(float((f_a[:, c0:c0 + 2] - f_i[:, c0:c0 + 2]).abs().mean()))
    assert np.mean(agree_inc) >= np.mean(agree_fresh) - 0.03, (agree_inc, agree_fresh)
    # 3 x 16 proxies per object sit closer to the query pixels than 16: the features move, but by less than the features themselves vary
    assert np.mean(diff_inc) <= max(3.0 * np.mean(diff_fresh), 0.05), (diff_inc, diff_fresh)


def test_bottleneck_golden(aoc, golden):
    """The decoder's residual block (gct.py:38-90) with the GCT gate and the GroupNorm + ReLU (+ residual) streams on the HIP library,
    against the output of the reference class."""
    g = golden("bottleneck_64_128")
    blk = aoc.gct.Bottleneck(64, 128).cuda()
    sd = {k[2:].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("p_")}
    blk.load_state_dict(sd)
    y = blk(dev(g["in_x"]))
    np.testing.assert_allclose(y.cpu().numpy(), g["out"], rtol=2e-5, atol=2e-5)
    t1 = blk._gn(blk.bn1, blk.conv1(blk.GCT1(dev(g["in_x"]))))
    np.testing.assert_allclose(t1.cpu().numpy(), g["stage1"], rtol=2e-5, atol=5e-6)
    # the fused GroupNorm + residual + ReLU stream by itself against torch
    x = torch.randn(3, 64, 17, 23, device="cuda")
    r = torch.randn_like(x)
    gn = torch.nn.GroupNorm(32, 64).cuda()
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.normal_()
    want = torch.relu(gn(x) + r)
    got = aoc.ops.groupnorm_relu(x, 32, gn.weight, gn.bias, gn.eps, r, True)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=2e-6)


def test_head_delta_equals_torch(aoc):
    """aoc_head_delta = torch.cat([head, px.sum(0, keepdim=True) - px], 1) (decoding_module.py:126-130)."""
    g = torch.Generator().manual_seed(5)
    for n_obj, D, C in [(4, 400, 512), (1, 400, 320), (9, 912, 128)]:
        head, px = torch.randn(n_obj, D, generator=g), torch.randn(n_obj, C, generator=g)
        got = aoc.ops.head_delta(head.cuda(), px.cuda()).cpu()
        want = torch.cat([head, px.sum(dim=0, keepdim=True) - px], dim=1)
        assert got.shape == want.shape and float((got - want).abs().max()) <= 2e-6


# ------------------------------------------------------------------------------------------ local matching, register-operand kernel
@pytest.mark.parametrize("C,h,w,mld,rate,down", [
    (100, 23, 37, [1, 3], 1, False),                 # 8 + 2R = 14 candidate columns: one group, odd map (partial 2 x 8 blocks)
    (100, 23, 37, [2, 4, 6, 8, 10, 12], 1, False),   # the model's windows
    (100, 21, 30, [15], 1, False),                   # R = 15: three groups
    (128, 23, 37, [2, 4, 6, 8, 10, 12], 1, False),   # C = 128: eight float4 pieces per lane, no tail channel
    (128, 40, 57, [3, 6, 9], 1, True),               # C = 128 with the 2x downsample
    (100, 26, 35, [4, 8, 12], 2, False),             # atrous 2: rows / columns at odd offsets contribute nothing
    (128, 26, 35, [3, 6, 12], 3, False),
])
def test_local_register_kernel_vs_oracle(aoc, C, h, w, mld, rate, down):
    """local_window_reg_kernel (C in {100, 128}) against the oracle over map sizes that leave partial workgroups, window sizes with one,
    two and three candidate groups, atrous rates, soft / overlapping / absent labels and a bias."""
    from oracle import matching as om
    rng = np.random.RandomState(C + h + len(mld) + rate)
    O = 4
    prev = torch.from_numpy((np.maximum(rng.randn(h, w, C), 0) * 0.3).astype(np.float32))
    cur = torch.from_numpy((0.8 * prev.numpy() + 0.2 * np.maximum(rng.randn(h, w, C), 0) * 0.3).astype(np.float32))
    ids = rng.randint(0, O + 1, size=(h, w))                        # id O: a pixel no object claims
    lab = np.stack([(ids == o) for o in range(O)], -1).astype(np.float32)
    lab[..., 1] = np.maximum(lab[..., 1], (rng.rand(h, w) < 0.1))    # overlapping labels
    lab *= np.where(rng.rand(h, w, 1) < 0.05, 0.5, 1.0).astype(np.float32)   # soft labels below the 0.9 threshold
    lab = torch.from_numpy(lab)
    bias = torch.tensor([0.2, -0.1, 0.0, 0.3])
    want = om.local_matching(prev, cur, lab, bias, mld, None, rate, False, down)
    got = aoc.matching.local_matching(prev.cuda(), cur.cuda(), lab.cuda(), bias.cuda().view(-1, 1, 1, 1), mld, None, rate, False, down, True)
    assert tuple(got.shape) == tuple(want.shape)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=ATOL)


@pytest.mark.parametrize("hw", [1, 3, 7, 1021, 121 * 213, 61 * 107])
def test_plane_mean_every_alignment(aoc, hw):
    """aoc_plane_mean with plane sizes that put the planes at every 4-byte phase of a 16-byte line (scalar head, 16-byte body, scalar tail)."""
    g = torch.Generator().manual_seed(hw)
    x = torch.randn(3, 5, hw, 1, generator=g)
    got = aoc.ops.plane_mean(x.cuda()).cpu()
    want = x.double().mean(dim=(2, 3)).float()
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-5, atol=2e-6)
