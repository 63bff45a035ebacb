"""aoc_frame_enqueue in every mode the reference's evaluation CLI / config can switch on (tools/eval_net_mm_rpa.py:9-35, configs/resnet101_aocnet.py:
MODEL_FLOAT16_MATCHING, MODEL_LOCAL_DOWNSAMPLE, TEST_LOCAL_ATROUS_RATE, TEST_GLOBAL_ATROUS_RATE) and with more than 16 objects:
  * EQUAL bit for bit to the Python orchestrator (hotpath.proto_mask_features) over a sequence whose pool grows;
  * within 5e-6 of the drop-in mirrors driven the way aocnet.py:196-337 drives the reference's functions with those switches (the mirrors are pinned to
    the reference's own outputs by the per-function goldens of tests/test_gpu_round2.py: atrous 2 / 3, use_float16, no down-sample); the dense and
    local channels, which take the same kernels on both sides, equal theirs exactly."""
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


def _mirrors_in_aocnet_order(aoc, mc, refs, labs, prev, prev_lab, cur, bias, seed):
    """aocnet.py:196-337 (eval branch) with the mirrors; returns the tensor handed to DynamicPreHead [O, n_ch, h, w] and the attention head."""
    m, att = aoc.matching, aoc.attention
    O = labs[0].shape[-1]
    f16, down = mc.MODEL_FLOAT16_MATCHING, mc.MODEL_LOCAL_DOWNSAMPLE
    gr, lr = mc.TEST_GLOBAL_ATROUS_RATE, mc.TEST_LOCAL_ATROUS_RATE
    mld = list(mc.MODEL_MULTI_LOCAL_DISTANCE)
    dis_bias = bias.view(-1, 1, 1, 1)
    nchw = lambda e: e.permute(2, 0, 1).unsqueeze(0).contiguous()
    np.random.seed(seed)
    g_fg = m.global_matching_for_eval(refs, cur, labs, 4, dis_bias, None, gr, f16)
    g_cl = m.global_matching_for_eval_cluster(refs, cur, labs, 4, dis_bias, None, gr, f16, 0,
                                              **({"cluster_num": list(mc.CLUSTER_LEVELS)} if mc.CLUSTER_LEVELS else {}))
    l_fg = m.local_matching(prev, cur, prev_lab, dis_bias, mld, None, lr, f16, down, True)
    labs_1hot = [l.permute(2, 0, 1).unsqueeze(1) for l in labs]
    prev_1hot = prev_lab.permute(2, 0, 1).unsqueeze(1)
    head, ref_pos, ref_neg, prev_pos, prev_neg = att.calculate_attention_head_for_eval_p_m(
        [nchw(e) for e in refs], labs_1hot, nchw(prev).expand((O, -1, -1, -1)), prev_1hot, epsilon=1e-5)
    g_px = m.global_matching_for_eval_proxy(ref_pos, cur, labs, 4, dis_bias, None, gr, f16)
    inst = torch.matmul(prev_lab, prev_pos)
    l_px = m.local_matching_proxy(inst, cur, prev_lab, dis_bias, mld, None, lr, f16, down, True)
    oc = lambda t: t.squeeze(0).permute(2, 3, 0, 1)
    if g_cl.shape[-1] == 1:                      # float16: the reference's cluster result has ONE channel of ones per level pair ... the model cats it as is;
        g_cl = g_cl.expand(-1, -1, -1, -1, 2 * len(mc.cluster_levels))       # the orchestrators write the constant into every cluster channel
    pre = torch.cat([oc(g_fg), oc(g_cl), oc(g_px), oc(l_fg), oc(l_px), prev_1hot], 1)
    if mc.MODEL_MATCHING_BACKGROUND:
        g_bg = m.foreground2background(oc(g_fg), O)
        l_bg = m.foreground2background(oc(l_fg).permute(0, 2, 3, 1).unsqueeze(1), O).permute(0, 4, 2, 3, 1).squeeze(-1)
        pre = torch.cat([pre, l_bg, g_bg], 1)
    return pre, head


CASES = {
    "float16": dict(f16=True),
    "no_downsample": dict(down=False),
    "local_atrous2": dict(lrate=2),
    "global_atrous2_odd_map": dict(grate=2, size=(25, 37)),
    "global3_local2_full_res": dict(grate=3, lrate=2, down=False),
    "float16_full_res_global2": dict(f16=True, down=False, grate=2),
    "eighteen_objects": dict(n_obj=18),
    "levels_8_16_32_global2": dict(levels=[8, 16, 32], grate=2),
    "no_background_float16": dict(f16=True, background=False),
}


@pytest.mark.parametrize("name", list(CASES))
def test_frame_call_modes(aoc, name):
    case = CASES[name]
    syn, hot, ops = aoc.synthetic, aoc.hotpath, aoc.ops
    h, w = case.get("size", (24, 40))
    O = case.get("n_obj", 3)
    cfg = syn.ClipConfig(name, h, w, O, 16, 100, 8, 3)
    clip = syn.make_clip(cfg, 33, frames=7)
    mc = hot.MatchingConfig(CLUSTER_LEVELS=case.get("levels"), MODEL_MATCHING_BACKGROUND=case.get("background", True), MEM_EVERY=3,
                            MODEL_FLOAT16_MATCHING=case.get("f16", False), MODEL_LOCAL_DOWNSAMPLE=case.get("down", True),
                            TEST_LOCAL_ATROUS_RATE=case.get("lrate", 1), TEST_GLOBAL_ATROUS_RATE=case.get("grate", 1))
    C = 100
    emb = torch.from_numpy(clip["emb"]).cuda()
    lab = torch.from_numpy(np.stack([syn.one_hot(l, O) for l in clip["lab"]])).cuda()
    bias = (torch.arange(O, dtype=torch.float32) * 0.07 - 0.2).cuda()
    assert hot.FrameRunner.supported(mc, C, O)
    runner = hot.FrameRunner(mc, h, w, C, O, capacity_frames=3, device=emb.device)
    side = torch.cuda.Stream()
    levels, kmax = mc.cluster_levels, max(mc.cluster_levels)
    dense_state = {}
    pool_ids = [0]
    for t in range(1, 7):
        ref_emb, ref_lab = emb[pool_ids].contiguous(), lab[pool_ids].contiguous()
        m_emb, m_lab = runner.match_pool(ref_emb, ref_lab)
        if mc.TEST_GLOBAL_ATROUS_RATE > 1:
            r = mc.TEST_GLOBAL_ATROUS_RATE
            assert torch.equal(m_emb, ref_emb[:, ::r, ::r]) and torch.equal(m_lab, ref_lab[:, ::r, ::r])
        seed = 500 + t
        if mc.MODEL_FLOAT16_MATCHING:
            ahead = hot.prepare_without_clustering(mc, m_emb, m_lab)
        else:
            counts = [int(m_lab[..., o].sum().item()) for o in range(O)]
            rows, _ = ops.kmeans_init_rows_draw(np.random.RandomState(seed), counts, levels, 1, kmax)       # numpy's stream, as the mirrors consume it
            ahead = hot.launch_cluster_proxies(mc, m_emb, m_lab, torch.from_numpy(rows[0]).cuda(), side)
        mp = (m_emb, m_lab) if mc.TEST_GLOBAL_ATROUS_RATE > 1 else None
        feat_c, head_c = runner(ref_emb, ref_lab, emb[t - 1], lab[t - 1], emb[t], bias, ahead, pool_key=len(pool_ids))
        feat_p, head_p, _ = hot.proto_mask_features(mc, ref_emb, ref_lab, emb[t - 1], lab[t - 1], emb[t], bias, cluster_ahead=ahead, dense_state=dense_state,
                                                    match_pool=mp)
        torch.cuda.synchronize()
        assert feat_c.shape == feat_p.shape == (O, mc.proto_channels, h, w)
        assert torch.equal(feat_c, feat_p), f"{name} frame {t}: frame call != Python orchestration (max {float((feat_c - feat_p).abs().max())})"
        assert torch.equal(head_c, head_p)
        pre, head = _mirrors_in_aocnet_order(aoc, mc, [e for e in ref_emb], [l for l in ref_lab], emb[t - 1], lab[t - 1], emb[t], bias, seed)
        ch = hot.channel_slices(mc)
        nl = len(mc.MODEL_MULTI_LOCAL_DISTANCE)
        err = (feat_c - pre).abs()
        assert float(err.max()) <= 5e-6, f"{name} frame {t}: frame call vs the mirrors in aocnet.py order: {float(err.max())}"
        assert torch.equal(feat_c[:, ch["local"]:ch["local"] + 2 * nl], pre[:, ch["local"]:ch["local"] + 2 * nl]), "local channels: same kernels on both sides"
        if O <= 16 or mc.MODEL_FLOAT16_MATCHING:
            assert torch.equal(feat_c[:, 0], pre[:, 0]), "dense channel: same kernel on both sides"
        np.testing.assert_allclose(head_c.cpu().numpy(), head.cpu().numpy(), rtol=1e-5, atol=1e-6)
        if t == 2 or t == 4:
            pool_ids.append(t)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float().cuda()


def _frame_for_golden(aoc, g, mode, kind, golden_loader):
    """One aoc_frame_enqueue call on a golden's inputs (the inputs the golden does not hold -- a previous frame for the global goldens, a pool for
    the local ones -- are seeded fillers: they only reach other channels).  Returns (features [O, n_ch, h, w], channel slices, config)."""
    hot, ops, syn = aoc.hotpath, aoc.ops, aoc.synthetic
    rng = np.random.RandomState(5)
    if kind == "local":
        cur, prev, prev_lab = _dev(g["in_query"]), _dev(g["in_prev"]), _dev(g["lab_onehot"])
        h, w, O = prev_lab.shape
        ref_emb, ref_lab = prev[None].contiguous(), prev_lab[None].contiguous()
        mld = [int(v) for v in g["mld"]]
    else:
        cur, ref_emb, ref_lab = _dev(g["in_query"]), _dev(g["in_ref"]), _dev(g["lab_onehot"])
        _, h, w, O = ref_lab.shape
        prev, prev_lab = ref_emb[-1].contiguous(), ref_lab[-1].contiguous()
        mld = [2, 4, 6, 8, 10, 12]
    mc = hot.MatchingConfig(MODEL_MULTI_LOCAL_DISTANCE=mld, MODEL_FLOAT16_MATCHING=mode.get("f16", False), MODEL_LOCAL_DOWNSAMPLE=mode.get("down", True),
                            TEST_LOCAL_ATROUS_RATE=mode.get("lrate", 1), TEST_GLOBAL_ATROUS_RATE=mode.get("grate", 1))
    bias = _dev(g["in_bias"])
    runner = hot.FrameRunner(mc, h, w, 100, O, capacity_frames=ref_emb.shape[0], device=cur.device)
    m_emb, m_lab = runner.match_pool(ref_emb, ref_lab)
    if mc.MODEL_FLOAT16_MATCHING:
        ahead = hot.prepare_without_clustering(mc, m_emb, m_lab)
    else:
        counts = [int(m_lab[..., o].sum().item()) for o in range(O)]
        seed = int(g["seed"]) if "seed" in g else 0
        rows, _ = ops.kmeans_init_rows_draw(np.random.RandomState(seed), counts, mc.cluster_levels, 1, max(mc.cluster_levels))
        ahead = hot.launch_cluster_proxies(mc, m_emb, m_lab, torch.from_numpy(rows[0]).cuda())
    feat, _ = runner(ref_emb, ref_lab, prev, prev_lab, cur, bias, ahead, pool_key=1)
    torch.cuda.synchronize()
    return feat, hot.channel_slices(mc), mc


@pytest.mark.parametrize("name,mode,kind", [
    ("dense_atrous2", dict(grate=2), "dense"), ("dense_atrous3_even", dict(grate=3), "dense"), ("cluster_atrous2", dict(grate=2), "cluster"),
    ("local_atrous2_down_O3", dict(lrate=2), "local"), ("local_atrous3_nodown_O3", dict(lrate=3, down=False), "local"),
    ("dense_fp16_R2_O3", dict(f16=True), "dense"), ("local_fp16_down_O3", dict(f16=True), "local"), ("local_atrous2_fp16_down_O3", dict(f16=True, lrate=2), "local")])
def test_reference_goldens_through_the_frame_call(aoc, golden, name, mode, kind):
    """The reference's OWN outputs for the atrous / float16 / full-resolution modes (tests/golden/make_golden_r2*.py ran the reference's functions
    with those arguments) against the matching channels of ONE aoc_frame_enqueue call on the same inputs."""
    g = golden(name)
    if "atrous_rate" in g:
        assert int(g["atrous_rate"]) == mode.get("grate", mode.get("lrate", 1))
    feat, ch, mc = _frame_for_golden(aoc, g, mode, kind, golden)
    want = torch.from_numpy(g["out"])[0].permute(2, 3, 0, 1).numpy()                # [1, h, w, O, F] -> [O, F, h, w]
    if kind == "dense":
        got = feat[:, ch["global_fg"]:ch["global_fg"] + 1]
    elif kind == "cluster":
        got = feat[:, ch["cluster"]:ch["cluster"] + 2]
    else:
        got = feat[:, ch["local"]:ch["local"] + len(mc.MODEL_MULTI_LOCAL_DISTANCE)]
    got = got.cpu().numpy()
    assert got.shape == want.shape
    if mode.get("f16"):
        diff = np.abs(got - want)                                                   # the float16-mode tolerance of tests/test_gpu_round2.py::_f16_close
        assert diff.max() <= 4e-3 and np.mean(diff <= 2e-6) >= 0.999, (diff.max(), np.mean(diff <= 2e-6))
    else:
        np.testing.assert_allclose(got, want, rtol=0, atol=5e-6)
