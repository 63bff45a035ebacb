"""aoc_frame_enqueue (ONE C call per frame, include/aoc_hip.h; hotpath.FrameRunner) against the Python-orchestrated path
(hotpath.proto_mask_features driving the individual entry points): the same kernels out of one persistent workspace per sequence, so the
proto-mask tensor and the attention head must be EQUAL bit for bit -- over a sequence whose pool grows (split records appended, pooled
reference heads and the dense plan rebuilt exactly when the pool changes), with the k-means chains on a side stream."""
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


def _init_rows_dev(syn, seed, counts, levels, n_obj):
    kmax = max(levels)
    rows = np.zeros((len(levels) * n_obj, kmax), np.int32)
    for li, k in enumerate(levels):
        for o, r in enumerate(syn.kmeans_init_rows(seed + li, counts, k)):
            if r is not None:
                rows[li * n_obj + o, :len(r)] = r
    return torch.from_numpy(rows).cuda()


@pytest.mark.parametrize("cfg_name,levels,background", [("tiny", None, True), ("tiny", [8, 16, 32], True), ("tiny", None, False), ("cfg1", None, True)])
def test_frame_call_equals_python_orchestration(aoc, cfg_name, levels, background):
    syn, hot = aoc.synthetic, aoc.hotpath
    cfg = syn.CONFIGS[cfg_name]
    T = 9 if cfg_name == "tiny" else 4
    clip = syn.make_clip(cfg, 21, frames=T)
    O, h, w, C = cfg.n_obj, cfg.h, cfg.w, cfg.c
    mc = hot.MatchingConfig(CLUSTER_LEVELS=levels, MODEL_MATCHING_BACKGROUND=background, MEM_EVERY=3)
    emb = torch.from_numpy(clip["emb"]).cuda()
    lab = torch.from_numpy(np.stack([syn.one_hot(l, O) for l in clip["lab"]])).cuda()
    bias = torch.tensor([0.25, -0.5, 0.125, 0.0, 0.3, -0.1][:O]).cuda()
    assert hot.FrameRunner.supported(mc, C, O)
    runner = hot.FrameRunner(mc, h, w, C, O, capacity_frames=4, device=emb.device)
    side = torch.cuda.Stream()
    dense_state = {}
    pool_ids = [0]
    for t in range(1, T):
        ref_emb, ref_lab = emb[pool_ids].contiguous(), lab[pool_ids].contiguous()
        counts = [int(ref_lab[..., o].sum().item()) for o in range(O)]
        init = _init_rows_dev(syn, 100 + t, counts, mc.cluster_levels, O)
        ahead = hot.launch_cluster_proxies(mc, ref_emb, ref_lab, init, side)
        feat_c, head_c = runner(ref_emb, ref_lab, emb[t - 1], lab[t - 1], emb[t], bias, ahead, pool_key=len(pool_ids))
        feat_c, head_c = feat_c.clone(), head_c.clone()
        feat_p, head_p, _ = hot.proto_mask_features(mc, ref_emb, ref_lab, emb[t - 1], lab[t - 1], emb[t], bias, cluster_ahead=ahead, dense_state=dense_state)
        torch.cuda.synchronize()
        assert feat_c.shape == feat_p.shape == (O, mc.proto_channels, h, w)
        assert torch.equal(feat_c, feat_p), f"frame {t}: proto-mask tensor differs (max {float((feat_c - feat_p).abs().max())})"
        assert torch.equal(head_c, head_p), f"frame {t}: attention head differs"
        if t % 3 == 0 and len(pool_ids) < 4:
            pool_ids.append(t)                      # the pool grows: new split records, new pooled heads, new dense plan
    assert len(pool_ids) >= 3 or cfg_name != "tiny"
    # a second sequence in the same workspace
    runner.reset()
    ref_emb, ref_lab = emb[[2]].contiguous(), lab[[2]].contiguous()
    counts = [int(ref_lab[..., o].sum().item()) for o in range(O)]
    ahead = hot.launch_cluster_proxies(mc, ref_emb, ref_lab, _init_rows_dev(syn, 7, counts, mc.cluster_levels, O), side)
    feat_c, head_c = runner(ref_emb, ref_lab, emb[2], lab[2], emb[3], bias, ahead, pool_key=1)
    feat_p, head_p, _ = hot.proto_mask_features(mc, ref_emb, ref_lab, emb[2], lab[2], emb[3], bias, cluster_ahead=ahead)
    assert torch.equal(feat_c, feat_p) and torch.equal(head_c, head_p)


def test_frame_call_vs_reference_golden(aoc, golden):
    """The one-call path against the tensors the reference's own before_seghead_process produced (tests/golden/make_golden_r4.py)."""
    from golden_cases import FRAME_CASES
    hot, ops = aoc.hotpath, aoc.ops
    for name in FRAME_CASES:
        g = golden(name)
        n_obj = int(g["n_obj"])
        h, w = g["in_cur"].shape[:2]
        mc = hot.MatchingConfig(MODEL_MATCHING_BACKGROUND=bool(g["background"]))
        dev = lambda a, dt=None: (torch.from_numpy(np.ascontiguousarray(a)) if dt is None else torch.from_numpy(np.ascontiguousarray(a)).to(dt)).cuda()
        ref_lab = torch.stack([ops.label_onehot_nearest(dev(l, torch.int32), h, w, n_obj) for l in g["ref_labels_full"]])
        prev_lab = ops.label_onehot_nearest(dev(g["prev_label_full"], torch.int32), h, w, n_obj)
        ref_emb = dev(g["in_ref"])
        # the initial rows scipy drew in the reference run (recorded per kmeans2 call, one per object)
        rows = np.zeros((n_obj, 16), np.int32)
        for i in range(int(g["km_calls"])):
            r = g[f"km{i}_rows"]
            rows[i, :len(r)] = r
        assert int(g["km_calls"]) == n_obj
        ahead = hot.launch_cluster_proxies(mc, ref_emb, ref_lab, torch.from_numpy(rows).cuda())
        runner = hot.FrameRunner(mc, h, w, 100, n_obj, ref_emb.shape[0], ref_emb.device)
        b = torch.full((n_obj,), float(g["fg_bias"]))
        b[0] = float(g["bg_bias"])
        feat, head = runner(ref_emb, ref_lab, dev(g["in_prev"]), prev_lab, dev(g["in_cur"]), b.cuda(), ahead, pool_key=ref_emb.shape[0])
        np.testing.assert_allclose(feat.cpu().numpy(), g["pre_to_cat"], rtol=0, atol=5e-6)
        np.testing.assert_allclose(head.cpu().numpy(), g["attention_head"], rtol=1e-5, atol=1e-6)


def test_gates_one_call_equals_the_modules(aoc):
    """aoc_gates_enqueue (CalibrationGates.forward_batched: the 10 IA gates and 4 conditioning blocks of CalibrationDecoding as ONE C call) against
    the module-by-module path (attention.IA_gate / conditioning_block mirrors): the same launches in the same order -> torch.equal."""
    hot = aoc.hotpath
    torch.manual_seed(5)
    mc = hot.MatchingConfig()
    gates = hot.CalibrationGates(mc).cuda()
    O, h, w = 3, 33, 45
    g = torch.Generator().manual_seed(9)
    acts = [torch.randn(O, c, hh, ww, generator=g).cuda() for (_, c, hh, ww, _) in gates.plan(h, w)]
    head = torch.randn(O, 400, generator=g).cuda()
    want = gates(acts, head)
    got = gates.forward_batched(acts, head)
    torch.cuda.synchronize()
    assert len(got) == len(want) == 14
    for name, a, b in zip([p[0] for p in gates.plan(h, w)], got, want):
        assert torch.equal(a, b), name
    # a second call re-uses descriptors and output buffers; another head gives another result
    got2 = [t.clone() for t in gates.forward_batched(acts, head * 0.5)]
    want2 = gates(acts, head * 0.5)
    for a, b in zip(got2, want2):
        assert torch.equal(a, b)


def test_gates_batch_follows_the_weights_storage(aoc):
    """the cached descriptors of forward_batched hold raw pointers of the module weights.  New weight STORAGE (load_state_dict(assign=True), a
    re-assigned parameter, module.to()) must rebuild them; an in-place update (the same storage) must simply be seen."""
    hot = aoc.hotpath
    torch.manual_seed(6)
    gates = hot.CalibrationGates(hot.MatchingConfig()).cuda()
    O, h, w = 2, 17, 23
    g = torch.Generator().manual_seed(10)
    acts = [torch.randn(O, c, hh, ww, generator=g).cuda() for (_, c, hh, ww, _) in gates.plan(h, w)]
    head = torch.randn(O, 400, generator=g).cuda()
    first = [t.clone() for t in gates.forward_batched(acts, head)]
    # in place: same storage, new values
    with torch.no_grad():
        gates.IA1.IA.weight.mul_(0.5)
    for a, b in zip(gates.forward_batched(acts, head), gates(acts, head)):
        assert torch.equal(a, b)
    # new storage for every parameter
    sd = {k: (v.clone() * 1.25) for k, v in gates.state_dict().items()}
    gates.load_state_dict(sd, assign=True)
    got = [t.clone() for t in gates.forward_batched(acts, head)]
    want = gates(acts, head)
    for a, b in zip(got, want):
        assert torch.equal(a, b), "the batch still reads the old weight storage"
    assert not torch.equal(got[0], first[0])
    # one re-assigned parameter
    gates.IA11.IA.bias = torch.nn.Parameter(torch.full_like(gates.IA11.IA.bias, 0.3))
    for a, b in zip(gates.forward_batched(acts, head), gates(acts, head)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("levels,F", [(None, 1), (None, 3), ([8, 16, 32], 2)])
def test_cluster_chain_one_call_equals_the_three_calls(aoc, levels, F):
    """aoc_cluster_chain_enqueue (round 5) against aoc_kmeans_replicate_levels + aoc_kmeans_segmented_rep + aoc_build_proxies + the table copies."""
    syn, hot, ops = aoc.synthetic, aoc.hotpath, aoc.ops
    cfg = syn.CONFIGS["tiny"]
    clip = syn.make_clip(cfg, 4, frames=3)
    O, C = cfg.n_obj, cfg.c
    mc = hot.MatchingConfig(CLUSTER_LEVELS=levels)
    lv, kmax = mc.cluster_levels, max(mc.cluster_levels)
    L = len(lv)
    emb = torch.from_numpy(clip["emb"][:2].copy()).cuda()
    lab = torch.from_numpy(np.stack([syn.one_hot(l, O) for l in clip["lab"][:2]])).cuda()
    pool = emb.reshape(-1, C)
    prep = ops.label_prep(lab.reshape(-1, O))
    counts = [int(lab[..., o].sum().item()) for o in range(O)]
    inits = [_init_rows_dev(syn, 40 + f, counts, lv, O) for f in range(F)]
    outs = hot.launch_cluster_proxies_batch(mc, emb, lab, inits)
    cap = prep.obj_rows.numel()
    rows_f, off_f, k_f = ops.kmeans_replicate_levels(prep.obj_rows, prep.obj_offsets, O, F * L, lv, rows_capacity=cap)
    init = torch.cat([r.reshape(L * O, kmax) for r in inits], dim=0)
    cen, labels, cnt = ops.kmeans_segmented(pool, rows_f, off_f, k_f, init, kmax, 20, rows_capacity=F * L * cap, n_rep=F * L)
    proxies, psq = ops.build_proxies(pool, prep.fg_rows, off_f, k_f, labels, cen)
    torch.cuda.synchronize()
    n_ad = L * O * 2 * kmax
    for f, out in enumerate(outs):
        sl = slice(f * L * O, (f + 1) * L * O)
        assert torch.equal(out.table[:n_ad], proxies[sl].reshape(-1, C)) and torch.equal(out.sqn[:n_ad], psq[sl].reshape(-1))
        assert torch.equal(out.aux["centroids"], cen[sl]) and torch.equal(out.aux["seg_k"], k_f[sl])
    ch = outs[0].aux["chain"]
    assert torch.equal(ch["labels"][:int(off_f[-1])], labels[:int(off_f[-1])]) and torch.equal(ch["cluster_counts"], cnt) and torch.equal(ch["seg_offsets"], off_f)


def test_cluster_chain_of_more_frames_than_one_descriptor_names(aoc):
    """Ten frames that see one pool state (MEM_EVERY > 9): the batch goes out as two chains over one label prep; every frame's table equals its own chain's."""
    syn, hot = aoc.synthetic, aoc.hotpath
    cfg = syn.CONFIGS["tiny"]
    clip = syn.make_clip(cfg, 9, frames=2)
    O = cfg.n_obj
    mc = hot.MatchingConfig()
    emb = torch.from_numpy(clip["emb"][:1].copy()).cuda()
    lab = torch.from_numpy(np.stack([syn.one_hot(l, O) for l in clip["lab"][:1]])).cuda()
    counts = [int(lab[..., o].sum().item()) for o in range(O)]
    inits = [_init_rows_dev(syn, 300 + f, counts, mc.cluster_levels, O) for f in range(10)]
    outs = hot.launch_cluster_proxies_batch(mc, emb, lab, inits)
    assert len(outs) == 10 and outs[9].prep is outs[0].prep
    for f in (0, 7, 8, 9):
        one = hot.launch_cluster_proxies(mc, emb, lab, inits[f])
        torch.cuda.synchronize()
        n_ad = len(mc.cluster_levels) * O * 2 * max(mc.cluster_levels)
        assert torch.equal(outs[f].table[:n_ad], one.table[:n_ad]) and torch.equal(outs[f].sqn[:n_ad], one.sqn[:n_ad])
