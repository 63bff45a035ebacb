"""Pins the CPU oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  The reference ships no tests of its own."""
import numpy as np
import pytest
import torch

import golden_cases  # tests/golden_cases.py
from oracle import calibration as ocal
from oracle import matching as om

T = torch.from_numpy
# float32 reassociation between the reference's chunked matmuls / unfold and the de-chunked
# oracle: features live in (-1, 1); observed differences are ~1e-6.
TOL = dict(rtol=0, atol=2e-6)


def _refs(g):
    return [T(e) for e in g["in_ref"]], [T(l.copy()) for l in g["lab_onehot"]]


def _ori(g):
    o = tuple(int(v) for v in g["ori_size"])
    return o if o[0] > 0 else None


CLUSTER_CASES = ["cluster_basic_R1_O3", "cluster_basic_R1_O3_mt", "cluster_R3_O4_bias", "cluster_orisize",
                 "cluster_sticky_empty_obj0", "cluster_small_first_obj", "cluster_small_mid_obj",
                 "cluster_duplicates_empty", "cluster_uncertain125", "cluster_all_unlabelled"]


@pytest.mark.parametrize("name", CLUSTER_CASES)
def test_cluster_path(golden, name):
    g = golden(name)
    refs, labs = _refs(g)
    np.random.seed(int(g["seed"]))           # the oracle draws init rows exactly as scipy does
    out, proxies = om.global_matching_for_eval_cluster(refs, T(g["in_query"]), labs, 4, T(g["in_bias"]),
                                                       _ori(g), return_proxies=True)
    assert tuple(out.shape) == g["out"].shape
    np.testing.assert_allclose(out.numpy(), g["out"], **TOL)
    # every kmeans2 call the reference made: same rows drawn, same labels, same code book (bit-exact)
    live = [p for p in (proxies or []) if p is not None]
    assert len(live) == int(g["km_calls"])
    for i, p in enumerate(live):
        assert p["k"] == int(g[f"km{i}_k"]) and len(p["labels"]) == int(g[f"km{i}_n"])
        assert np.array_equal(p["init_rows"], g[f"km{i}_rows"])
        assert np.array_equal(p["labels"], g[f"km{i}_labels"])
        assert np.array_equal(p["centroid"].numpy(), g[f"km{i}_centroid"])


def test_cluster_kmeans_iteration_trace(golden):
    """Labels after iterations 1, 2 and 20 of every recorded kmeans2 call."""
    from oracle import kmeans as okm
    g = golden("cluster_R3_O4_bias")
    ref = T(g["in_ref"]).reshape(-1, 100)
    lab = T(g["lab_onehot"]).reshape(-1, 4)
    keep = lab.sum(1) > 0.9
    ref, lab = ref[keep].numpy(), lab[keep].numpy()
    for i in range(int(g["km_calls"])):
        x = ref[lab[:, i] > 0.9]
        _, _, _, tr = okm.kmeans2_matrix(x, x[g[f"km{i}_rows"]], 20, trace=True)
        assert np.array_equal(tr[0], g[f"km{i}_labels_it1"])
        assert np.array_equal(tr[1], g[f"km{i}_labels_it2"])
        assert np.array_equal(tr[19], g[f"km{i}_labels"])


def test_cluster_special_values(golden):
    g = golden("cluster_sticky_empty_obj0")
    assert np.all(g["out"] == 1.0)                      # sticky K = 0 (AEM:268)
    g = golden("cluster_all_unlabelled")
    assert g["out"].shape[-1] == 1 and np.all(g["out"] == 1.0)   # early-out has last dim 1 (AEM:588-589)
    g = golden("cluster_small_first_obj")
    assert [int(g[f"km{i}_k"]) for i in range(int(g["km_calls"]))] == [3, 3, 3]


@pytest.mark.parametrize("name", ["dense_R1_O3", "dense_R2_O4_bias_unc", "dense_orisize", "dense_all_unlabelled"])
def test_dense_path(golden, name):
    g = golden(name)
    refs, labs = _refs(g)
    out = om.global_matching_for_eval(refs, T(g["in_query"]), labs, 4, T(g["in_bias"]), _ori(g))
    assert tuple(out.shape) == g["out"].shape
    np.testing.assert_allclose(out.numpy(), g["out"], **TOL)


def test_dense_train_twin(golden):
    g = golden("dense_train_twin")
    out = om.global_matching(T(g["in_ref"][0]), T(g["in_query"]), T(g["lab_onehot"][0].copy()), 3, T(g["in_bias"]))
    np.testing.assert_allclose(out.numpy(), g["out"], **TOL)


@pytest.mark.parametrize("name,fn", [("proxy_eval_O3", "global_matching_for_eval_proxy"), ("proxy_train_O3", "global_matching_proxy")])
def test_proxy_path(golden, name, fn):
    g = golden(name)
    labs = T(g["lab_onehot"][0].copy())
    if fn == "global_matching_proxy":
        out = om.global_matching_proxy(T(g["in_proxies"]), T(g["in_query"]), labs, 3, T(g["in_bias"]))
    else:
        out = om.global_matching_for_eval_proxy(T(g["in_proxies"]), T(g["in_query"]), [labs], 4, T(g["in_bias"]))
    np.testing.assert_allclose(out.numpy(), g["out"], **TOL)


@pytest.mark.parametrize("name", ["local_down_O3", "local_down_O4_bias", "local_nodown_O3",
                                  "local_down_odd_C36", "local_proxy_down_O3"])
def test_local_path(golden, name):
    g = golden(name)
    out = om.local_matching(T(g["in_prev"]), T(g["in_query"]), T(g["lab_onehot"].copy()), T(g["in_bias"]),
                            [int(v) for v in g["mld"]], None, 1, False, bool(g["down"]))
    assert tuple(out.shape) == g["out"].shape
    np.testing.assert_allclose(out.numpy(), g["out"], **TOL)


@pytest.mark.parametrize("name,n", [("fg2bg_O4", 4), ("fg2bg_O1", 1)])
def test_fg2bg(golden, name, n):
    g = golden(name)
    assert np.array_equal(om.foreground2background(T(g["in_dis"]), n).numpy(), g["out"])


def _onehot(lab, n):
    return (T(lab.astype(np.int64)).unsqueeze(0) == torch.arange(n).view(-1, 1, 1)).float().unsqueeze(1)


def test_attention_head(golden):
    g = golden("attention_head_eval_p_m")
    n = int(g["n_obj"])
    ref_e = [T(e).permute(2, 0, 1).unsqueeze(0) for e in g["in_ref"]]
    ref_l = [_onehot(l, n) for l in g["lab_ref"]]
    prev_e = T(g["in_prev"]).permute(2, 0, 1).unsqueeze(0).expand(n, -1, -1, -1)
    outs = ocal.attention_head_for_eval_p_m(ref_e, ref_l, prev_e, _onehot(g["lab_prev"], n))
    for o, key in zip(outs, ["total", "ref_pos", "ref_neg", "prev_pos", "prev_neg"]):
        np.testing.assert_allclose(o.numpy(), g[key], rtol=1e-6, atol=1e-7)
    # the training twin (ATT:134-153) with one reference frame gives the same head
    g2 = golden("attention_head_train_p_m")
    o2 = ocal.attention_head_for_eval_p_m([T(g2["in_ref"][0]).permute(2, 0, 1).unsqueeze(0)], [_onehot(g2["lab_ref"][0], n)],
                                          T(g2["in_prev"]).permute(2, 0, 1).unsqueeze(0).expand(n, -1, -1, -1),
                                          _onehot(g2["lab_prev"], n))[0]
    np.testing.assert_allclose(o2.numpy(), g2["total"], rtol=1e-6, atol=1e-7)


def test_ia_gate(golden):
    g = golden("ia_gate")
    y = ocal.ia_gate(T(g["in_x"]), T(g["in_head"]), T(g["in_w"]), T(g["in_b"]))
    np.testing.assert_allclose(y.numpy(), g["out"], rtol=1e-6, atol=1e-7)


def test_conditioning_layer(golden):
    g = golden("conditioning_layer_4d")
    z = T(g["in_z"])
    s, thr, mask, gap = ocal.conditioning_gate_stats(z, T(g["in_phi_w"]), T(g["in_phi_b"]), float(g["beta"]))
    np.testing.assert_allclose(s.numpy(), g["scores"], rtol=1e-5, atol=1e-6)
    k = int(0.3 * z.shape[2] * z.shape[3])
    assert np.array_equal(mask.sum(1).numpy(), g["mask_count"]) and np.all(g["mask_count"] == k - 1)
    out = ocal.conditioning_layer(z, T(g["in_phi_w"]), T(g["in_phi_b"]), T(g["in_mlp_w"]), T(g["in_mlp_b"]), float(g["beta"]))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-5, atol=1e-6)


def test_conditioning_block_repair_is_consistent():
    """a13 is unpinned by the reference (not executable).  Check the documented repair: the block
    equals x * (1 + tanh(mlp([CL_1(x), mlp_2(x_delta), mlp_3(head)]))) built from pinned pieces."""
    rng = np.random.RandomState(0)
    n, c, p = 3, 8, 6
    x = T(rng.randn(n, c, 5, 7).astype(np.float32))
    head = T(rng.randn(n, p).astype(np.float32))
    w = {k: T(rng.randn(*s).astype(np.float32) * 0.2) for k, s in {
        "CL_1.phi_w": (c,), "CL_1.phi_b": (1,), "CL_1.mlp_w": (c, c), "CL_1.mlp_b": (c,),
        "CL_2.mlp_w": (c, c), "CL_2.mlp_b": (c,), "CL_3.mlp_w": (p, p), "CL_3.mlp_b": (p,),
        "mlp_w": (c, 2 * c + p), "mlp_b": (c,)}.items()}
    y = ocal.conditioning_block(x, head, w)
    px = x.mean(dim=(2, 3))
    delta = px.sum(0, keepdim=True) - px
    c1 = ocal.conditioning_layer(x, w["CL_1.phi_w"], w["CL_1.phi_b"], w["CL_1.mlp_w"], w["CL_1.mlp_b"])
    a = 1 + torch.tanh(torch.cat([c1, delta @ w["CL_2.mlp_w"].t() + w["CL_2.mlp_b"],
                                  head @ w["CL_3.mlp_w"].t() + w["CL_3.mlp_b"]], 1) @ w["mlp_w"].t() + w["mlp_b"])
    np.testing.assert_allclose(y.numpy(), (a[:, :, None, None] * x).numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("case", golden_cases.BLOCK_CASES)
def test_conditioning_block_against_the_reference_forward(golden, case):
    """a13, round 6: the reference's OWN conditioning_block.forward (CLB:66-86), executed unmodified with CL_1 / CL_2 / CL_3 / mlp_layer injected as
    module globals (CL_1 = the reference layer, CL_2 / CL_3 = the documented vector repair; tests/golden/make_golden_r6.py).  Pins CLB:68-69, CLB:72
    through the real layer and CLB:81-84; what stays pinned by the restatement alone is the repair v -> mlp(v) itself."""
    g = golden(case)
    w, _ = golden_cases.block_weights(g)
    x, head = T(g["in_x"].astype(np.float32)), T(g["in_head"].astype(np.float32))
    c1 = ocal.conditioning_layer(x, w["CL_1.phi_w"], w["CL_1.phi_b"], w["CL_1.mlp_w"], w["CL_1.mlp_b"], float(g["beta"]))
    np.testing.assert_allclose(c1.numpy(), g["cl1"], rtol=1e-5, atol=1e-6)
    y = ocal.conditioning_block(x, head, w, float(g["beta"]))
    np.testing.assert_allclose(y.numpy(), g["out"], rtol=1e-5, atol=2e-6)


def test_shannon_entropy_matches_reference_output(golden):
    """oracle/eval_loop.py::cal_shannon_entropy against the reference's own function (tests/golden/make_golden_entropy.py)."""
    from oracle import eval_loop as oe
    g = golden("shannon_entropy")
    got = oe.cal_shannon_entropy(torch.from_numpy(g["preds"]))
    np.testing.assert_array_equal(got.numpy(), g["uncertainty"])
    # the frame decision with every label seen reduces to the same map
    _, _, unc = oe.frame_decision(torch.from_numpy(g["preds"]), list(range(g["preds"].shape[1])))
    np.testing.assert_array_equal(unc.numpy(), g["uncertainty"][0, 0])


# ------------------------------------------------------------------------------------------ round-2 fixtures (make_golden_r2.py)
def _check_km_calls(g, proxies_per_level):
    """Every kmeans2 call the reference made, in call order (level-major, objects in order)."""
    live = [p for lv in proxies_per_level for p in (lv or []) if p is not None]
    assert len(live) == int(g["km_calls"])
    for i, p in enumerate(live):
        assert p["k"] == int(g[f"km{i}_k"]) and len(p["labels"]) == int(g[f"km{i}_n"])
        assert np.array_equal(p["init_rows"], g[f"km{i}_rows"])
        assert np.array_equal(p["labels"], g[f"km{i}_labels"])
        assert np.array_equal(p["centroid"].numpy(), g[f"km{i}_centroid"])


@pytest.mark.parametrize("name", ["cluster_K8_R1_O3", "cluster_K32_R2_O4", "cluster_levels_8_16_32_R2_O4", "cluster_levels_small_obj"])
def test_cluster_levels(golden, name):
    """cluster_num != 16 (AEM:232) and the multi-level K in {8, 16, 32} configuration (BASELINE.json configs[2])."""
    g = golden(name)
    refs, labs = _refs(g)
    levels = [int(v) for v in g["levels"]]
    np.random.seed(int(g["seed"]))
    out, prox = om.global_matching_for_eval_cluster(refs, T(g["in_query"]), labs, 4, T(g["in_bias"]), None, return_proxies=True,
                                                    cluster_num=levels if len(levels) > 1 else levels[0])
    assert tuple(out.shape) == g["out"].shape and out.shape[-1] == 2 * len(levels)
    np.testing.assert_allclose(out.numpy(), g["out"], **TOL)
    _check_km_calls(g, prox if len(levels) > 1 else [prox])
    if name == "cluster_levels_small_obj":      # object 1 has 12 pixels: K sticks at 12 from there on for the levels 16 and 32
        assert [int(g[f"km{i}_k"]) for i in range(int(g["km_calls"]))] == [8, 8, 8, 16, 12, 12, 32, 12, 12]


@pytest.mark.parametrize("name,fn", [("cluster_atrous2", "cluster"), ("cluster_atrous2_objpix", "cluster"), ("dense_atrous2", "dense"),
                                     ("dense_atrous2_objpix", "dense"), ("dense_atrous3_even", "dense")])
def test_atrous_pool_flattening(golden, name, fn):
    """AEM:513-579 / 715-787 with atrous_rate > 1, with and without atrous_obj_pixel_num."""
    g = golden(name)
    refs, labs = _refs(g)
    rate, objpix = int(g["atrous_rate"]), int(g["atrous_obj_pixel_num"])
    np.random.seed(int(g["seed"]))
    if fn == "cluster":
        out, prox = om.global_matching_for_eval_cluster(refs, T(g["in_query"]), labs, 4, T(g["in_bias"]), None, rate, False, objpix, return_proxies=True)
        _check_km_calls(g, [prox])
    else:
        out = om.global_matching_for_eval(refs, T(g["in_query"]), labs, 4, T(g["in_bias"]), None, rate, False, objpix)
    assert tuple(out.shape) == g["out"].shape
    np.testing.assert_allclose(out.numpy(), g["out"], **TOL)


@pytest.mark.parametrize("name", ["dense_train_twin_atrous2", "cluster_train_twin", "cluster_train_twin_atrous2", "cluster_train_twin_unlabelled"])
def test_training_twins(golden, name):
    """AEM:616-685 with atrous label masking, and matching.py:1324 global_matching_cluster2 (two-channel early-out)."""
    g = golden(name)
    rate, objpix = int(g["atrous_rate"]), int(g["atrous_obj_pixel_num"])
    np.random.seed(int(g["seed"]))
    args = (T(g["in_ref"][0]), T(g["in_query"]), T(g["lab_onehot"][0].copy()), 3, T(g["in_bias"]), None, rate, False, objpix)
    out = om.global_matching(*args) if name.startswith("dense") else om.global_matching_cluster(*args)
    assert tuple(out.shape) == g["out"].shape
    np.testing.assert_allclose(out.numpy(), g["out"], **TOL)
    if name == "cluster_train_twin_unlabelled":
        assert g["out"].shape[-1] == 2 and np.all(g["out"] == 1.0)          # AEM:455-456


@pytest.mark.parametrize("name,mode,relu", [("gct_l2", "l2", False), ("gct_l1", "l1", False), ("gct_l1_relu", "l1", True)])
def test_gct_vs_reference_class(golden, name, mode, relu):
    g = golden(name)
    sh = (1, -1, 1, 1)
    y = ocal.gct_forward(T(g["in_x"]), T(g["in_alpha"]).view(sh), T(g["in_gamma"]).view(sh), T(g["in_beta"]).view(sh), float(g["eps"]), mode, relu)
    np.testing.assert_allclose(y.numpy(), g["out"], rtol=1e-6, atol=1e-7)


def test_bottleneck_prehead_ia_logit_vs_reference(golden):
    g = golden("bottleneck_64_128")
    p = {k[2:].replace("__", "."): T(v) for k, v in g.items() if k.startswith("p_")}
    y, s1 = ocal.bottleneck(T(g["in_x"]), p)
    np.testing.assert_allclose(s1.numpy(), g["stage1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(y.numpy(), g["out"], rtol=1e-5, atol=2e-6)
    g = golden("dynamic_prehead")
    y = ocal.dynamic_prehead(T(g["in_x"]), T(g["in_conv_w"]), T(g["in_conv_b"]), T(g["in_gn_w"]), T(g["in_gn_b"]), int(g["groups"]), float(g["eps"]))
    np.testing.assert_allclose(y.numpy(), g["out"], rtol=1e-5, atol=1e-6)
    g = golden("ia_logit")
    y = ocal.ia_logit(T(g["in_x"]), T(g["in_head"]), T(g["in_w"]), T(g["in_b"]))
    np.testing.assert_allclose(y.numpy(), g["out"], rtol=1e-5, atol=1e-6)
    aug = ocal.augment_background_logit(y[:, :1], y[:, :1] * 0.5 - 0.1)
    np.testing.assert_allclose(aug.numpy(), g["aug"], rtol=1e-5, atol=1e-6)


def fullsize_inputs(g):
    """The full-size cfg1 frame pair is regenerated from its seed; the fixture stores the SHA-256 of the inputs."""
    import hashlib
    from aoc_amd import synthetic as syn
    cfg = syn.CONFIGS["cfg1"]
    d = syn.make_clip(cfg, seed=int(g["clip_seed"]), frames=2)
    sha = np.frombuffer(hashlib.sha256(d["emb"].tobytes() + d["lab"].tobytes()).digest(), np.uint8)
    assert np.array_equal(sha, g["in_sha256"]), "synthetic clip generator no longer reproduces the recorded inputs"
    return cfg, d


def check_fullsize(g, key, out, atol_sub, atol_f16=1.5e-3):
    a = out.reshape(-1, *out.shape[-2:])                               # [hw, O, F]
    np.testing.assert_allclose(a[::7], g[f"{key}_sub"], rtol=0, atol=atol_sub)                   # exact float32 sub-sample
    np.testing.assert_allclose(a, g[f"{key}_f16"].reshape(a.shape), rtol=0, atol=atol_f16)       # every pixel, float16-compressed
    np.testing.assert_allclose(a.astype(np.float64).sum(0), g[f"{key}_sum"], rtol=0, atol=atol_sub * a.shape[0] * 0.05 + 1e-3)


def test_fullsize_cfg1_oracle(golden):
    """one full-size cfg1 frame (121x213, O = 2, K = 16) of the reference: cluster (bit-exact k-means), dense, local."""
    import os
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    with np.load(os.path.join(os.path.dirname(__file__), "golden", "fullsize_cfg1.npz")) as z:
        g = {k: z[k] for k in z.files}
    cfg, d = fullsize_inputs(g)
    e0, e1 = T(d["emb"][0]), T(d["emb"][1])
    from aoc_amd import synthetic as syn
    l0 = T(syn.one_hot(d["lab"][0], cfg.n_obj))
    bias = torch.zeros(cfg.n_obj)
    np.random.seed(int(g["seed"]))
    out, prox = om.global_matching_for_eval_cluster([e0], e1, [l0], 4, bias, return_proxies=True)
    g2 = dict(g)
    for i in range(int(g["km_calls"])):
        g2[f"km{i}_labels"] = g[f"km{i}_labels"].astype(np.int32)
    _check_km_calls(g2, [prox])
    check_fullsize(g, "cluster", out.numpy()[0], 2e-6)
    check_fullsize(g, "local", om.local_matching(e0, e1, l0, bias, [2, 4, 6, 8, 10, 12]).numpy()[0], 2e-6)
    # dense: every 7th query pixel (the [m, n] distance matrix of the full frame is 2.6 GB)
    q = e1.reshape(-1, cfg.c)[::7]
    dn = om.proto_transform(om.nearest_neighbor_features_per_object(e0.reshape(-1, cfg.c), q, l0.reshape(-1, cfg.n_obj)).squeeze(-1), bias.view(1, -1))
    np.testing.assert_allclose(dn.numpy()[:, :, None], g["dense_sub"], rtol=0, atol=2e-6)


# ------------------------------------------------------------------------------------------ use_float16=True (reference .half() paths)
def _f16_close(got, want, frac_exact=1.0, atol=2e-6):
    """float16-mode outputs: distances are float16 TENSORS in the reference; the oracle restates its steps with the same torch-CPU operations
    (same shapes, same memory formats), so it reproduces the reference's own `.half()` outputs on every element."""
    diff = np.abs(got - want)
    assert diff.max() <= atol, diff.max()
    assert np.mean(diff <= 2e-6) >= frac_exact, np.mean(diff <= 2e-6)


def test_float16_dense_and_local_vs_reference_half_path(golden):
    g = golden("dense_fp16_R2_O3")
    refs, labs = _refs(g)
    out = om.global_matching_for_eval(refs, T(g["in_query"]), labs, 4, T(g["in_bias"]), None, 1, True, 0)
    assert out.dtype == torch.float32 and tuple(out.shape) == g["out"].shape
    _f16_close(out.numpy(), g["out"])
    g = golden("local_fp16_down_O3")
    out = om.local_matching(T(g["in_prev"]), T(g["in_query"]), T(g["lab_onehot"].copy()), T(g["in_bias"]), [int(v) for v in g["mld"]], None, 1, True,
                            bool(g["down"]))
    assert out.dtype == torch.float32
    _f16_close(out.numpy(), g["out"])
    g = golden("cluster_fp16_R2_O3")                    # scipy rejects float16 -> bare except -> 5e4 -> exactly 1.0 (AEM:275-286)
    assert g["out"].shape[-1] == 2 and np.all(g["out"] == 1.0) and int(g["km_calls"]) == 0


@pytest.mark.parametrize("name", ["local_atrous2_down_O3", "local_atrous3_nodown_O3", "local_atrous2_fp16_down_O3"])
def test_local_atrous(golden, name):
    """AEM:949-959 (window offsets are multiples of atrous_rate up to max - max % rate), AEM:1039 (nested radii // rate)."""
    g = golden(name)
    out = om.local_matching(T(g["in_prev"]), T(g["in_query"]), T(g["lab_onehot"].copy()), T(g["in_bias"]), [int(v) for v in g["mld"]], None,
                            int(g["atrous_rate"]), bool(g["float16"]), bool(g["down"]))
    assert tuple(out.shape) == g["out"].shape
    np.testing.assert_allclose(out.numpy(), g["out"], **TOL)


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 4: the per-frame orchestration (f-1) and the eval-loop bookkeeping (f-2) against the reference ITSELF
# (tests/golden/make_golden_r4.py: AOCNet.before_seghead_process and Evaluator.evaluating run unmodified on mock objects).
from golden_cases import FRAME_CASES, check_eval_loop, frame_inputs  # noqa: E402  (tests/golden_cases.py)


@pytest.mark.parametrize("name", FRAME_CASES)
def test_frame_orchestration_vs_reference(golden, name):
    """oracle/hotpath.py against aocnet.py:114-372 executed by the reference: label prep from full-resolution maps (nearest resize,
    label 125, an object absent from a reference frame), bias assembly, call order under ONE RandomState stream, the channel order of
    the 24 (17 without MODEL_MATCHING_BACKGROUND) proto-mask channels, DynamicPreHead, the attention head."""
    from oracle import hotpath as oh
    g = golden(name)
    np.random.seed(int(g["seed"]))
    pre, head, out = oh.before_seghead_process_eval(**frame_inputs(g))
    assert tuple(pre.shape) == g["pre_to_cat"].shape == (int(g["n_obj"]), 24 if g["background"] else 17, *g["in_cur"].shape[:2])
    np.testing.assert_allclose(pre.numpy(), g["pre_to_cat"], **TOL)
    np.testing.assert_allclose(head.numpy(), g["attention_head"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out.numpy(), g["prehead_out"], rtol=1e-4, atol=2e-5)
    # the previous-frame mask channel IS what the decoder is handed as to_cat_previous_frame (aocnet.py:152,356,367)
    assert np.array_equal(pre[:, 16:17].numpy(), g["seghead_prev_mask"])


def test_frame_goldens_hold_what_they_claim(golden):
    g = golden("frame_R2_O3_absent_unc")
    assert (g["ref_labels_full"] == 125).sum() > 0 and not (g["ref_labels_full"][1] == 2).any() and (g["ref_labels_full"][0] == 2).any()
    g = golden("frame_R3_O4_bias")
    assert g["in_ref"].shape[0] == 3 and int(g["km_calls"]) == 4 and float(g["bg_bias"]) != float(g["fg_bias"]) != 0.0
    assert golden("frame_R2_O3_nobg")["pre_to_cat"].shape[1] == 17


@pytest.mark.parametrize("name", ["eval_loop_join_obj3", "eval_loop_mem3"])
def test_eval_loop_bookkeeping_vs_reference(golden, name):
    """oracle/eval_loop.py::MemoryPolicy against Evaluator.evaluating executed by the reference (recording mock model)."""
    from oracle import eval_loop as oe
    g = golden(name)
    check_eval_loop(g, oe.MemoryPolicy(mem_every=int(g["mem_every"]), unc_ratio=float(g["unc_ratio"])))


def test_eval_loop_goldens_hold_what_they_claim(golden):
    g = golden("eval_loop_join_obj3")
    assert int(g["n_frames"]) >= 12 and int(g["mem_every"]) == 5
    assert g["f13_ref_frames"].tolist() == [0, 5, 7, 10]                       # MEM_EVERY frames and the frame that carries ground truth
    assert (g["f13_ref_masks"] == 125).any()                                   # uncertain pixels reach the model as label 125
    assert not (g["saved_labels"][:6] == 3).any() and (g["saved_labels"][6] == 3).any()      # object 3 joins at frame 7
    assert not (g["saved_labels"] == 4).any() and g["probs"].shape[1] == 5                   # channel 4: a label never seen in any ground truth
