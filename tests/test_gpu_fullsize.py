"""GPU parity at BASELINE.json's full sizes (cfg2: 121x213 maps, C=100, O=4, K=16; cfg4-like: O=9, K=64) and for the
wider configurations (multi-level K, many objects).  Where the CPU oracle finishes in seconds the comparison is direct;
the dense branch (133 GFLOP per frame on the CPU) is compared on a sub-sample of query pixels plus size-independent
properties (monotonicity under pool growth, invariance to a permutation of the reference pixels)."""
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
    aoc_amd._lib.lib()
    return aoc_amd


def _clip(aoc, name, frames, seed=3, **over):
    syn = aoc.synthetic
    cfg = syn.CONFIGS[name]
    if over:
        cfg = syn.ClipConfig(**{**cfg.__dict__, **over})
    clip = syn.make_clip(cfg, seed, frames=frames)
    emb = torch.from_numpy(clip["emb"])
    lab = torch.from_numpy(np.stack([syn.one_hot(l, cfg.n_obj) for l in clip["lab"]]))
    return cfg, clip, emb, lab


def test_kmeans_full_size_bit_exact(aoc):
    """cfg2, two reference frames (51 546 rows): labels, counts and code books equal the oracle (== scipy) bit for bit."""
    from oracle import kmeans as okm
    cfg, clip, emb, lab = _clip(aoc, "cfg2", 6)
    pool = emb[[0, 5]].reshape(-1, cfg.c)
    labels = lab[[0, 5]].reshape(-1, cfg.n_obj)
    np.random.seed(11)
    cp = aoc.matching.cluster_proxies(pool.cuda(), labels.cuda())
    offs = cp["prep"].obj_offsets.cpu().numpy()
    got_lab, got_cen, got_cnt = cp["labels"].cpu().numpy(), cp["centroids"].cpu().numpy(), cp["cluster_counts"].cpu().numpy()
    ids = np.concatenate([clip["lab"][0].reshape(-1), clip["lab"][5].reshape(-1)])
    for o in range(cfg.n_obj):
        x = pool.numpy()[ids == o]
        cb, l, cnt = okm.kmeans2_matrix(x, x[cp["init_rows"][o]], 20)
        assert np.array_equal(got_lab[offs[o]:offs[o + 1]], l), f"object {o}"
        assert np.array_equal(got_cen[o], cb) and np.array_equal(got_cnt[o], cnt)


def test_cluster_and_local_full_size_vs_oracle(aoc):
    from oracle import matching as om
    cfg, clip, emb, lab = _clip(aoc, "cfg2", 3)
    bias = torch.tensor([0.1, -0.1, 0.2, 0.0])
    rows = aoc.synthetic.kmeans_init_rows(5, [int((clip["lab"][0] == o).sum()) for o in range(cfg.n_obj)], 16)
    want = om.global_matching_for_eval_cluster([emb[0]], emb[2], [lab[0]], 4, bias, init_rows=rows)
    got = aoc.matching.global_matching_for_eval_cluster([emb[0].cuda()], emb[2].cuda(), [lab[0].cuda()], 4, bias.cuda().view(-1, 1, 1, 1),
                                                        None, 1, False, 0, init_rows=rows)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=ATOL)
    mld = [2, 4, 6, 8, 10, 12]
    want = om.local_matching(emb[1], emb[2], lab[1], bias, mld)
    got = aoc.matching.local_matching(emb[1].cuda(), emb[2].cuda(), lab[1].cuda(), bias.cuda().view(-1, 1, 1, 1), mld, None, 1, False, True, True)
    assert tuple(got.shape) == (1, 121, 213, 4, 6)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=ATOL)


def test_dense_full_size_subsample_and_properties(aoc):
    from oracle import matching as om
    cfg, clip, emb, lab = _clip(aoc, "cfg2", 7)
    bias = torch.zeros(cfg.n_obj)
    refs, labs = [emb[0].cuda(), emb[5].cuda()], [lab[0].cuda(), lab[5].cuda()]
    got2 = aoc.matching.global_matching_for_eval(refs, emb[6].cuda(), labs, 4, bias.cuda(), None, 1, False, 0)[0, :, :, :, 0]
    # (1) sub-sample of query pixels against the oracle
    sel = torch.arange(0, cfg.h * cfg.w, 97)
    ref_flat = emb[[0, 5]].reshape(-1, cfg.c)
    lab_flat = lab[[0, 5]].reshape(-1, cfg.n_obj)
    want = om.proto_transform(om.nearest_neighbor_features_per_object(ref_flat, emb[6].reshape(-1, cfg.c)[sel], lab_flat).squeeze(-1), bias.view(1, -1))
    np.testing.assert_allclose(got2.reshape(-1, cfg.n_obj).cpu()[sel].numpy(), want.numpy(), rtol=0, atol=ATOL)
    # (2) a larger pool can only bring a pixel closer (min over a superset)
    got1 = aoc.matching.global_matching_for_eval(refs[:1], emb[6].cuda(), labs[:1], 4, bias.cuda(), None, 1, False, 0)[0, :, :, :, 0]
    assert bool((got2 <= got1 + 1e-7).all())
    # (3) the order of the reference frames (a permutation of the pool rows) does not matter
    got2r = aoc.matching.global_matching_for_eval(refs[::-1], emb[6].cuda(), labs[::-1], 4, bias.cuda(), None, 1, False, 0)[0, :, :, :, 0]
    assert torch.equal(got2, got2r)
    # (4) matching a frame against itself: every pixel finds itself at distance ~0 for its own object
    self_m = aoc.matching.global_matching_for_eval(refs[:1], emb[0].cuda(), labs[:1], 4, bias.cuda(), None, 1, False, 0)[0, :, :, :, 0]
    own = torch.gather(self_m, 2, torch.from_numpy(clip["lab"][0]).long().cuda().unsqueeze(-1)).squeeze(-1)
    assert float(own.abs().max()) < 2e-5


@pytest.mark.parametrize("n_obj,k,h,w", [(9, 64, 45, 81), (6, 32, 37, 66), (6, 8, 37, 66)])
def test_many_objects_and_large_k(aoc, n_obj, k, h, w):
    """cfg3 / cfg4 style: up to 9 objects and 64 proxies per object (multi-tile sets, O > 8 dense path)."""
    from oracle import matching as om
    cfg, clip, emb, lab = _clip(aoc, "tiny", 3, h=h, w=w, n_obj=n_obj, k=k)
    bias = torch.linspace(-0.2, 0.2, n_obj)
    # dense
    want = om.global_matching_for_eval([emb[0]], emb[2], [lab[0]], 4, bias)
    got = aoc.matching.global_matching_for_eval([emb[0].cuda()], emb[2].cuda(), [lab[0].cuda()], 4, bias.cuda(), None, 1, False, 0)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=ATOL)
    # cluster proxies with K != 16 through the orchestrated (sync-free) path vs the oracle's building blocks
    from oracle import kmeans as okm
    counts = [int((clip["lab"][0] == o).sum()) for o in range(n_obj)]
    rows = aoc.synthetic.kmeans_init_rows(4, counts, k)
    pool, labels = emb[0].reshape(-1, cfg.c), lab[0].reshape(-1, n_obj)
    cp = aoc.matching.cluster_proxies(pool.cuda(), labels.cuda(), k, rows)
    offs = cp["prep"].obj_offsets.cpu().numpy()
    kk = k
    for o in range(n_obj):
        kk = min(kk, counts[o])
        if kk == 0:
            continue
        x = pool.numpy()[clip["lab"][0].reshape(-1) == o]
        cb, l, cnt = okm.kmeans2_matrix(x, x[rows[o][:kk]], 20)
        assert np.array_equal(cp["labels"].cpu().numpy()[offs[o]:offs[o + 1]], l)
        assert np.array_equal(cp["centroids"].cpu().numpy()[o, :kk], cb)
    # whole proto-mask tensor with K proxies per object
    from aoc_amd import hotpath
    from oracle import hotpath as ohot
    mc = hotpath.MatchingConfig(CLUSTER_NUM=k)
    if k == 16:
        return
    init = np.zeros((n_obj, k), np.int32)
    for o, r in enumerate(rows):
        if r is not None:
            init[o, :len(r)] = r
    feat, head, aux = hotpath.proto_mask_features(mc, emb[:1].cuda(), lab[:1].cuda(), emb[1].cuda(), lab[1].cuda(), emb[2].cuda(), bias.cuda(),
                                                  cluster_state=dict(init_rows=torch.from_numpy(init).cuda()))
    assert tuple(feat.shape) == (n_obj, 24, h, w) and bool(torch.isfinite(feat).all())
    # cluster channels against the oracle's proxy correlation with the same (bit-identical) proxies
    q = emb[2].reshape(-1, cfg.c)
    qsq = q.pow(2).sum(1)
    P, N = aux["cluster"]["proxies"].cpu(), aux["cluster"]["proxy_sqnorm"].cpu()
    for o in range(n_obj):
        for f in range(2):
            live = torch.isfinite(N[o, f])
            if not bool(live.any()):
                want_ch = torch.full((h * w,), 5e4)
            else:
                c = P[o, f][live]
                want_ch = om.flattened_pairwise_distances(c, c.pow(2).sum(1), q, qsq).min(1)[0]
            want_ch = om.proto_transform(want_ch, bias[o])
            np.testing.assert_allclose(feat[o, 1 + f].reshape(-1).cpu().numpy(), want_ch.numpy(), rtol=0, atol=ATOL)


def test_ragged_and_degenerate_inputs(aoc):
    """m not a multiple of 16, a single reference pixel per object, an object absent from the pool."""
    from oracle import matching as om
    rng = np.random.RandomState(2)
    h, w, c, o = 7, 9, 100, 3
    ref = torch.from_numpy((np.maximum(rng.randn(h, w, c), 0) * 0.3).astype(np.float32))
    q = torch.from_numpy((np.maximum(rng.randn(h, w, c), 0) * 0.3).astype(np.float32))
    ids = np.zeros((h, w), np.int64)
    ids[3, 4] = 1                                   # object 1: one pixel; object 2: absent
    lab = torch.from_numpy((ids[..., None] == np.arange(o)).astype(np.float32))
    bias = torch.tensor([0.0, 0.3, -0.3])
    want = om.global_matching_for_eval([ref], q, [lab], 4, bias)
    got = aoc.matching.global_matching_for_eval([ref.cuda()], q.cuda(), [lab.cuda()], 4, bias.cuda(), None, 1, False, 0)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=ATOL)
    np.random.seed(1)
    want = om.global_matching_for_eval_cluster([ref], q, [lab], 4, bias)
    np.random.seed(1)
    got = aoc.matching.global_matching_for_eval_cluster([ref.cuda()], q.cuda(), [lab.cuda()], 4, bias.cuda(), None, 1, False, 0)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=ATOL)
    want = om.local_matching(ref, q, lab, bias, [1, 2, 3], None, 1, False, False)
    got = aoc.matching.local_matching(ref.cuda(), q.cuda(), lab.cuda(), bias.cuda(), [1, 2, 3], None, 1, False, False, True)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=ATOL)


def test_kmeans_exact_on_signed_and_constant_data(aoc):
    """The integer-domain folding only applies to non-negative data; anything else must fall back to the literal serial
    additions and still be bit-exact: signed values, dead (all-zero) channels, constant channels, huge dynamic range."""
    from oracle import kmeans as okm
    rng = np.random.RandomState(9)
    n, c, k = 5000, 100, 16
    x = rng.randn(n, c).astype(np.float32)                      # signed
    x[:, 5] = 0.0                                               # dead channel
    x[:, 6] = 0.25                                              # constant channel (ties galore)
    x[:, 7] = np.abs(x[:, 7]) * np.float32(1e-20)               # tiny values
    x[:, 8] = np.abs(x[:, 8]) * np.float32(1e6)                 # large values
    x[:, 9] = np.float32(0.5) * rng.randint(0, 5, n)            # half-integers: exact ties
    rows = np.arange(n, dtype=np.int32)
    init = rng.permutation(n)[:k].astype(np.int32)[None]
    cen, lab, cnt = aoc.ops.kmeans_segmented(torch.from_numpy(x).cuda(), torch.from_numpy(rows).cuda(),
                                              torch.tensor([0, n], dtype=torch.int32).cuda(), torch.tensor([k], dtype=torch.int32).cuda(),
                                              torch.from_numpy(init).cuda(), k, 20)
    cb, l, ct = okm.kmeans2_matrix(x, x[init[0]], 20)
    assert np.array_equal(lab.cpu().numpy(), l) and np.array_equal(cen.cpu().numpy()[0], cb) and np.array_equal(cnt.cpu().numpy()[0], ct)


def test_cfg4_full_size_kmeans_and_dense(aoc):
    """cfg4 (181x321 maps, O=9, K=64): bit-exact k-means for the two smallest objects' segments and for the background,
    dense branch on a sub-sample of query pixels, whole proto-mask tensor finite."""
    from oracle import kmeans as okm
    from oracle import matching as om
    cfg, clip, emb, lab = _clip(aoc, "cfg4", 3)
    O, K = cfg.n_obj, cfg.k
    counts = [int((clip["lab"][0] == o).sum()) for o in range(O)]
    rows = aoc.synthetic.kmeans_init_rows(21, counts, K)
    pool, labels = emb[0].reshape(-1, cfg.c), lab[0].reshape(-1, O)
    cp = aoc.matching.cluster_proxies(pool.cuda(), labels.cuda(), K, rows)
    offs = cp["prep"].obj_offsets.cpu().numpy()
    got_lab, got_cen = cp["labels"].cpu().numpy(), cp["centroids"].cpu().numpy()
    kk = K
    order = np.argsort(counts)
    check = {int(order[0]), int(order[1]), int(order[-1])}
    for o in range(O):
        kk = min(kk, counts[o])
        if o not in check or kk == 0:
            continue
        x = pool.numpy()[clip["lab"][0].reshape(-1) == o]
        cb, l, cnt = okm.kmeans2_matrix(x, x[rows[o][:kk]], 20)
        assert np.array_equal(got_lab[offs[o]:offs[o + 1]], l), f"object {o}"
        assert np.array_equal(got_cen[o, :kk], cb)
    bias = torch.linspace(-0.2, 0.2, O)
    got = aoc.matching.global_matching_for_eval([emb[0].cuda()], emb[2].cuda(), [lab[0].cuda()], 4, bias.cuda(), None, 1, False, 0)[0, :, :, :, 0]
    sel = torch.arange(0, cfg.h * cfg.w, 211)
    want = om.proto_transform(om.nearest_neighbor_features_per_object(pool, emb[2].reshape(-1, cfg.c)[sel], labels).squeeze(-1), bias.view(1, -1))
    np.testing.assert_allclose(got.reshape(-1, O).cpu()[sel].numpy(), want.numpy(), rtol=0, atol=ATOL)
    from aoc_amd import hotpath
    mc = hotpath.MatchingConfig(CLUSTER_NUM=K)
    feat, head, _ = hotpath.proto_mask_features(mc, emb[:1].cuda(), lab[:1].cuda(), emb[1].cuda(), lab[1].cuda(), emb[2].cuda(), bias.cuda(), init_rows=rows)
    assert tuple(feat.shape) == (O, 24, cfg.h, cfg.w) and bool(torch.isfinite(feat).all())


def test_kmeans_bit_exact_with_the_single_pass_tail():
    """AOC_KM_FUSED=1 (km_chunk_scanfold_kernel: chunk sums, look-back and folds in one launch) is a developer switch of the DEVELOPMENT build
    (libaoc_hip_dev.so, AOC_LIB_VARIANT=dev), read once per process: the bit-exactness tests of the k-means pipeline are re-run in a child
    process with that library and the switch set."""
    import os, subprocess, sys
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    here = os.path.dirname(os.path.abspath(__file__))
    import aoc_amd
    if not os.path.exists(aoc_amd._lib.DEV_SO):
        pytest.skip("development build not present (make DEV=1)")
    env = dict(os.environ, AOC_KM_FUSED="1", AOC_LIB_VARIANT="dev")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(here, "test_gpu_fullsize.py"), os.path.join(here, "test_gpu_parity.py"),
                        "-k", "kmeans and not single_pass and not persistent"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_kmeans_bit_exact_on_a_segment_beyond_the_inline_prediction_limit(aoc):
    """ONE object of 460 000 rows (cfg4's background at a dozen pool frames): more than 409 600 rows per segment, so iteration 0 takes the
    separate binade-prediction launch; iterations 1..19 fold their ~860 tail chunks in the binades the previous iteration recorded (round 5).
    Labels, counts and the code book equal the oracle (== scipy) bit for bit."""
    from oracle import kmeans as okm
    rng = np.random.RandomState(77)
    n, C, K = 460_000, 100, 16
    x = (np.maximum(rng.randn(n, C), 0.0) * 0.3).astype(np.float32)
    x[:, 5] = 0.0                                                   # a feature that is identically zero: sums stay 0 (the literal fallback)
    lab = np.zeros((n, 2), np.float32)
    lab[:, 0] = 1.0
    lab[-50:, 0], lab[-50:, 1] = 0.0, 1.0                           # a second, tiny object
    init = np.zeros((2, K), np.int32)
    init[0] = rng.permutation(n - 50)[:K]
    init[1] = rng.permutation(50)[:K]
    pool, labels = torch.from_numpy(x).cuda(), torch.from_numpy(lab).cuda()
    prep = aoc.ops.label_prep(labels)
    seg_k = aoc.ops.kmeans_plan(prep.counts, 2, K)
    cen, got_lab, got_cnt = aoc.ops.kmeans_segmented(pool, prep.obj_rows, prep.obj_offsets, seg_k, torch.from_numpy(init).cuda(), K, 20, rows_capacity=prep.obj_rows.numel())
    offs = prep.obj_offsets.cpu().numpy()
    for o, rows in enumerate((np.arange(n - 50), np.arange(n - 50, n))):
        cb, l, cnt = okm.kmeans2_matrix(x[rows], x[rows][init[o]], 20)
        assert np.array_equal(got_lab.cpu().numpy()[offs[o]:offs[o + 1]], l), f"object {o}: labels"
        assert np.array_equal(got_cnt.cpu().numpy()[o], cnt) and np.array_equal(cen.cpu().numpy()[o], cb), f"object {o}: code book"
