"""Round 4: the product against goldens produced by the reference's OWN orchestration code (tests/golden/make_golden_r4.py):
``AOCNet.before_seghead_process`` (aocnet.py:114-372) and ``Evaluator.evaluating`` (eval_manager_mm.py:160-394) executed unmodified."""
import numpy as np
import pytest
import torch

from golden_cases import FRAME_CASES, check_eval_loop

pytestmark = pytest.mark.gpu
FEAT_TOL = dict(rtol=0, atol=5e-6)


@pytest.fixture(scope="module")
def aoc():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import aoc_amd
    aoc_amd._lib.lib()
    return aoc_amd


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).cuda()


def _prehead(aoc, g):
    m = aoc.hotpath.DynamicPreHead(in_dim=g["pre_to_cat"].shape[1], embed_dim=64).cuda()
    with torch.no_grad():
        m.conv.weight.copy_(_dev(g["prehead_conv_w"]))
        m.conv.bias.copy_(_dev(g["prehead_conv_b"]))
        m.bn.weight.copy_(_dev(g["prehead_gn_w"]))
        m.bn.bias.copy_(_dev(g["prehead_gn_b"]))
    assert m.bn.num_groups == int(g["prehead_groups"]) and abs(m.bn.eps - float(g["prehead_eps"])) < 1e-12
    return m


def _dis_bias(g, n_obj):
    """aocnet.py:143-146: cat(bg_bias, fg_bias.expand(gt_ids)) shaped [O,1,1,1]."""
    b = torch.full((n_obj,), float(g["fg_bias"]))
    b[0] = float(g["bg_bias"])
    return b.view(n_obj, 1, 1, 1).cuda()


@pytest.mark.parametrize("name", FRAME_CASES)
def test_orchestrated_frame_vs_reference(aoc, golden, name):
    """aoc_label_onehot_nearest -> hotpath.proto_mask_features (every kernel writes its channel slice) -> DynamicPreHead against the
    tensors aocnet.py handed to dynamic_prehead / dynamic_seghead."""
    g = golden(name)
    hot, ops = aoc.hotpath, aoc.ops
    n_obj = int(g["n_obj"])
    h, w = g["in_cur"].shape[:2]
    mc = hot.MatchingConfig(MODEL_MATCHING_BACKGROUND=bool(g["background"]))
    ref_lab = torch.stack([ops.label_onehot_nearest(_dev(l, torch.int32), h, w, n_obj) for l in g["ref_labels_full"]])
    prev_lab = ops.label_onehot_nearest(_dev(g["prev_label_full"], torch.int32), h, w, n_obj)
    np.random.seed(int(g["seed"]))
    feat, head, _ = hot.proto_mask_features(mc, _dev(g["in_ref"]), ref_lab, _dev(g["in_prev"]), prev_lab, _dev(g["in_cur"]), _dis_bias(g, n_obj))
    assert tuple(feat.shape) == g["pre_to_cat"].shape
    np.testing.assert_allclose(feat.cpu().numpy(), g["pre_to_cat"], **FEAT_TOL)
    np.testing.assert_allclose(head.cpu().numpy(), g["attention_head"], rtol=1e-5, atol=1e-6)
    y = _prehead(aoc, g)(feat)
    np.testing.assert_allclose(y.cpu().numpy(), g["prehead_out"], rtol=1e-4, atol=3e-5)
    # aocnet.py:362: the decoder's input = cat(current embedding per object, prehead output), in one launch
    cur = _dev(g["in_cur"])
    both = _prehead(aoc, g)(feat, cur)
    want = np.concatenate([np.broadcast_to(g["in_cur"].transpose(2, 0, 1)[None], (n_obj, 100, h, w)), g["prehead_out"]], 1)
    np.testing.assert_allclose(both.cpu().numpy(), want, rtol=1e-4, atol=3e-5)


@pytest.mark.parametrize("name", FRAME_CASES)
def test_reference_api_in_aocnet_order_vs_reference(aoc, golden, name):
    """The drop-in mirrors (same names, argument order and layouts as the reference's modules), driven the way aocnet.py drives the
    reference's functions: [1, C, h, w] embeddings, full-resolution integer label maps, permuted (non-contiguous) views, the results
    permuted and concatenated by torch -- no orchestrator of this repo in between."""
    g = golden(name)
    m, att = aoc.matching, aoc.attention
    F = torch.nn.functional
    n_obj = int(g["n_obj"])
    h, w = g["in_cur"].shape[:2]
    mld = [2, 4, 6, 8, 10, 12]
    nchw = lambda e: _dev(e).permute(2, 0, 1).unsqueeze(0).contiguous()
    ref_emb = [nchw(e) for e in g["in_ref"]]
    prev_emb, cur_emb = nchw(g["in_prev"]), nchw(g["in_cur"])
    H, W = g["prev_label_full"].shape
    ref_full = [_dev(l.astype(np.int64)).view(1, 1, H, W) for l in g["ref_labels_full"]]
    prev_full = _dev(g["prev_label_full"].astype(np.int64)).view(1, 1, H, W)
    ids = torch.arange(0, n_obj, device="cuda").int().view(-1, 1, 1, 1)
    dis_bias = _dis_bias(g, n_obj)

    small = lambda l: F.interpolate(l.float(), size=(h, w), mode="nearest").int()
    prev_1hot = (small(prev_full)[0] == ids).float()                      # [O,1,h,w]
    prev_hwo = prev_1hot.squeeze(1).permute(1, 2, 0)
    q = cur_emb[0].permute(1, 2, 0)
    p = prev_emb[0].permute(1, 2, 0)
    refs, labs, labs_1hot = [], [], []
    for e, l in zip(ref_emb, ref_full):
        refs.append(e[0].permute(1, 2, 0))
        one = (small(l)[0] == ids).float()
        labs_1hot.append(one)
        labs.append(one.squeeze(1).permute(1, 2, 0))

    np.random.seed(int(g["seed"]))
    g_fg = m.global_matching_for_eval(all_reference_embeddings=refs, query_embeddings=q, all_reference_labels=labs, n_chunks=4, dis_bias=dis_bias,
                                      atrous_rate=1, use_float16=False)
    g_cl = m.global_matching_for_eval_cluster(all_reference_embeddings=refs, query_embeddings=q, all_reference_labels=labs, n_chunks=4,
                                              dis_bias=dis_bias, atrous_rate=1, use_float16=False)
    l_fg = m.local_matching(prev_frame_embedding=p, query_embedding=q, prev_frame_labels=prev_hwo, multi_local_distance=mld, dis_bias=dis_bias,
                            use_float16=False, atrous_rate=1, allow_downsample=True, allow_parallel=True)
    head, ref_pos, ref_neg, prev_pos, prev_neg = att.calculate_attention_head_for_eval_p_m(
        ref_emb, labs_1hot, prev_emb[0].unsqueeze(0).expand((n_obj, -1, -1, -1)), prev_1hot, epsilon=1e-5)
    g_px = m.global_matching_for_eval_proxy(all_reference_embeddings=ref_pos, query_embeddings=q, all_reference_labels=labs, n_chunks=4,
                                            dis_bias=dis_bias, atrous_rate=1, use_float16=False)
    inst = torch.matmul(prev_hwo, prev_pos)
    l_px = m.local_matching_proxy(prev_frame_embedding=inst, query_embedding=q, prev_frame_labels=prev_hwo, multi_local_distance=mld,
                                  dis_bias=dis_bias, use_float16=False, atrous_rate=1, allow_downsample=True, allow_parallel=True)
    oc = lambda t: t.squeeze(0).permute(2, 3, 0, 1)
    parts = [oc(g_fg), oc(g_cl), oc(g_px), oc(l_fg), oc(l_px), prev_1hot]
    pre = torch.cat(parts, 1)
    if bool(g["background"]):
        g_bg = m.foreground2background(oc(g_fg), n_obj)
        l_bg = m.foreground2background(oc(l_fg).permute(0, 2, 3, 1).unsqueeze(1), n_obj).permute(0, 4, 2, 3, 1).squeeze(-1)
        pre = torch.cat([pre, l_bg, g_bg], 1)
    np.testing.assert_allclose(pre.cpu().numpy(), g["pre_to_cat"], **FEAT_TOL)
    np.testing.assert_allclose(head.cpu().numpy(), g["attention_head"], rtol=1e-5, atol=1e-6)
    y = _prehead(aoc, g)(pre)
    np.testing.assert_allclose(y.cpu().numpy(), g["prehead_out"], rtol=1e-4, atol=3e-5)


@pytest.mark.parametrize("name", ["eval_loop_join_obj3", "eval_loop_mem3"])
def test_memory_policy_vs_reference_loop(aoc, golden, name):
    """eval_loop.MemoryPolicy (aoc_confident_labels on the device + the list bookkeeping) against what Evaluator.evaluating handed its
    model every frame: pool membership, every confident reference mask incl. 125, the previous mask, the saved label maps."""
    g = golden(name)
    check_eval_loop(g, aoc.eval_loop.MemoryPolicy(mem_every=int(g["mem_every"]), unc_ratio=float(g["unc_ratio"])), to_dev=lambda t: t.cuda())


def test_release_library_ignores_developer_switches():
    """The wrong-result developer switches of the development build (AOC_DENSE_DEBUG, AOC_CORR_DEBUG, ...) set in the environment of a
    process that loads the RELEASE library change nothing: the golden parity tests of the kernels they used to reach pass unchanged."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, AOC_DENSE_DEBUG="7", AOC_CORR_DEBUG="6", AOC_KR_DEBUG="1", AOC_KM_SUM="scan", AOC_KM_ASSIGN="valu", AOC_LOCAL_KERNEL="block",
               AOC_DENSE_CUS="8", AOC_CORR_CUS="8", AOC_KM_CHAIN="persistent")
    env.pop("AOC_LIB_VARIANT", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(here, "test_gpu_round4.py"), "-k", "orchestrated_frame"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
