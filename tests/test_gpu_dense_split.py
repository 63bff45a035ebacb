"""Split-fp16 dense matching (aoc_dense_match_min_split) against the exact-fp32 kernels and the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ATOL = 5e-6          # on proto-mask outputs (sigmoid slope <= 0.5)
ATOL_RAW = 1e-5      # on raw squared distances of magnitude O(1..10)


@pytest.fixture(scope="module")
def aoc():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import aoc_amd
    aoc_amd._lib.lib()
    return aoc_amd


def _dense(aoc, q, pool, labels, bias, precision, transform=True):
    ops = aoc.ops
    prep = ops.label_prep(labels.cuda())
    m, O = q.shape[0], labels.shape[1]
    out = torch.empty(O, m, device="cuda")
    ops.dense_match(q.cuda(), pool.cuda(), prep, None if bias is None else bias.cuda(), out, 1, m, transform, precision=precision)
    return out.cpu()


def _data(seed, n, m, c=100, o=4, scale=0.3, signed=False):
    rng = np.random.RandomState(seed)
    f = (lambda a: a) if signed else (lambda a: np.maximum(a, 0))
    pool = torch.from_numpy((f(rng.randn(n, c)) * scale).astype(np.float32))
    q = torch.from_numpy((f(rng.randn(m, c)) * scale).astype(np.float32))
    ids = rng.randint(0, o, n)
    lab = torch.from_numpy((ids[:, None] == np.arange(o)).astype(np.float32))
    return q, pool, lab


@pytest.mark.parametrize("n,m,o,c,signed", [(5000, 3000, 4, 100, False), (777, 1001, 3, 100, True), (4097, 513, 9, 100, False),
                                             (1500, 700, 2, 64, False), (33, 65, 16, 100, False), (40000, 2000, 4, 100, False)])
def test_split_matches_fp32_and_oracle(aoc, n, m, o, c, signed):
    from oracle import matching as om
    q, pool, lab = _data(n + m, n, m, c, o, signed=signed)
    bias = torch.linspace(-0.3, 0.3, o)
    raw_s = _dense(aoc, q, pool, lab, None, "split", transform=False)
    raw_f = _dense(aoc, q, pool, lab, None, "fp32", transform=False)
    assert float((raw_s - raw_f).abs().max()) < ATOL_RAW
    got = _dense(aoc, q, pool, lab, bias, "split")
    if n * m <= 2e7:
        want = om.proto_transform(om.nearest_neighbor_features_per_object(pool, q, lab).squeeze(-1), bias.view(1, -1))
        np.testing.assert_allclose(got.t().numpy(), want.numpy(), rtol=0, atol=ATOL)
    # the nearest reference pixel (argmin) agrees wherever the two best candidates are not within rounding of each other
    assert float((_dense(aoc, q, pool, lab, bias, "fp32") - got).abs().max()) < ATOL


def test_split_absent_object_and_single_pixels(aoc):
    from oracle import matching as om
    q, pool, lab = _data(5, 300, 200, o=5)
    lab[:, 3] = 0                                   # object 3 absent: 5e4 + nearest other pixel (AEM:84-88)
    lab[:, 4] = 0
    lab[lab.sum(1) == 0, 0] = 1
    lab[7] = 0
    lab[7, 4] = 1                                   # object 4: one pixel
    bias = torch.zeros(5)
    got = _dense(aoc, q, pool, lab, bias, "split", transform=False)
    want = om.nearest_neighbor_features_per_object(pool, q, lab).squeeze(-1)
    np.testing.assert_allclose(got.t().numpy(), want.numpy(), rtol=1e-6, atol=ATOL_RAW)


def test_take_over_on_soft_labels_is_bit_identical_to_fp32(aoc):
    """Labels that are not one-hot (soft / multi-object / unlabeled-but-kept rows) hand the call to the fp32 kernels."""
    q, pool, lab = _data(9, 2000, 600, o=3)
    lab[::7] = torch.tensor([0.5, 0.5, 0.0])        # neither wrong for 0 nor 1, right for none
    lab[1::11] = torch.tensor([1.0, 1.0, 0.0])      # right for two objects
    a = _dense(aoc, q, pool, lab, torch.zeros(3), "split")
    b = _dense(aoc, q, pool, lab, torch.zeros(3), "fp32")
    assert torch.equal(a, b)


def test_take_over_on_fp16_overflow_is_bit_identical_to_fp32(aoc):
    q, pool, lab = _data(10, 1000, 300, o=2)
    pool[17, 3] = 80.0                              # 80 * 1024 > 65504
    a = _dense(aoc, q, pool, lab, torch.zeros(2), "split")
    b = _dense(aoc, q, pool, lab, torch.zeros(2), "fp32")
    assert torch.equal(a, b)
    q2, pool2, lab2 = _data(11, 1000, 300, o=2, scale=8.0)   # |x|^2 beyond the norm-slot range
    a = _dense(aoc, q2, pool2, lab2, torch.zeros(2), "split", transform=False)
    b = _dense(aoc, q2, pool2, lab2, torch.zeros(2), "fp32", transform=False)
    assert torch.equal(a, b)


def test_tiny_and_huge_dynamic_range(aoc):
    """Values far below the fp16 normal range (after scaling) must not lose more than fp32 rounding."""
    from oracle import matching as om
    q, pool, lab = _data(12, 900, 400, o=3)
    pool[:, :10] *= 1e-4
    q[:, :10] *= 1e-4
    pool[:, 10:20] *= 10.0
    q[:, 10:20] *= 10.0
    got = _dense(aoc, q, pool, lab, None, "split", transform=False).t().double()
    want = om.nearest_neighbor_features_per_object(pool, q, lab).squeeze(-1).double()
    # |q|^2 + |r|^2 is O(300) here, so (|q|^2 + |r|^2) - 2 q.r cancels ~5 bits: judge both fp32 paths against float64
    p64, q64 = pool.double(), q.double()
    d64 = (q64.pow(2).sum(1)[:, None] + p64.pow(2).sum(1)[None, :]) - 2.0 * q64 @ p64.t()
    truth = torch.stack([d64[:, lab[:, o] > 0.5].min(1)[0] for o in range(3)], 1)
    err_split, err_fp32 = float((got - truth).abs().max()), float((want - truth).abs().max())
    assert err_split <= max(2.0 * err_fp32, ATOL_RAW), (err_split, err_fp32)


def test_nothing_labelled(aoc):
    q, pool, lab = _data(13, 200, 100, o=2)
    lab[:] = 0
    assert torch.equal(_dense(aoc, q, pool, lab, torch.zeros(2), "split"), torch.ones(2, 100))


def test_pool_cache_across_frames(aoc):
    """hotpath dense_state: records of earlier frames are reused; result equals the uncached call."""
    syn, hot = aoc.synthetic, aoc.hotpath
    cfg = syn.CONFIGS["tiny"]
    clip = syn.make_clip(cfg, 4, frames=5)
    emb = torch.from_numpy(clip["emb"]).cuda()
    lab = torch.from_numpy(np.stack([syn.one_hot(l, cfg.n_obj) for l in clip["lab"]])).cuda()
    mc = hot.MatchingConfig()
    bias = torch.zeros(cfg.n_obj, device="cuda")
    state = {"capacity_frames": 4}
    for R in (1, 2, 3):
        rows = syn.kmeans_init_rows(R, [int((clip["lab"][:R] == o).sum()) for o in range(cfg.n_obj)], 16)
        f1, _, _ = hot.proto_mask_features(mc, emb[:R], lab[:R], emb[3], lab[3], emb[4], bias, init_rows=rows, dense_state=state)
        f2, _, _ = hot.proto_mask_features(mc, emb[:R], lab[:R], emb[3], lab[3], emb[4], bias, init_rows=rows)
        f3, _, _ = hot.proto_mask_features(mc, emb[:R], lab[:R], emb[3], lab[3], emb[4], bias, init_rows=rows, dense_precision="fp32")
        assert torch.equal(f1, f2)
        assert float((f1 - f3).abs().max()) < ATOL
    assert state["frames"] == 3


def test_cluster_proxies_launched_ahead_equal_inline(aoc):
    """hotpath.launch_cluster_proxies (k-means chain enqueued ahead on a side stream) == the inline cluster branch."""
    syn, hot = aoc.synthetic, aoc.hotpath
    cfg = syn.CONFIGS["tiny"]
    clip = syn.make_clip(cfg, 6, frames=5)
    emb = torch.from_numpy(clip["emb"]).cuda()
    lab = torch.from_numpy(np.stack([syn.one_hot(l, cfg.n_obj) for l in clip["lab"]])).cuda()
    mc = hot.MatchingConfig()
    bias = torch.zeros(cfg.n_obj, device="cuda")
    side = torch.cuda.Stream()
    for R in (1, 3):
        rows = syn.kmeans_init_rows(R, [int((clip["lab"][:R] == o).sum()) for o in range(cfg.n_obj)], 16)
        init = np.zeros((cfg.n_obj, 16), np.int32)
        for o, r in enumerate(rows):
            if r is not None:
                init[o, :len(r)] = r
        init = torch.from_numpy(init).cuda()
        ev = torch.cuda.Event()
        ev.record()
        ahead = hot.launch_cluster_proxies(mc, emb[:R], lab[:R], init, side, wait_event=ev)
        f1, h1, _ = hot.proto_mask_features(mc, emb[:R], lab[:R], emb[3], lab[3], emb[4], bias, cluster_ahead=ahead)
        f2, h2, _ = hot.proto_mask_features(mc, emb[:R], lab[:R], emb[3], lab[3], emb[4], bias, cluster_state=dict(init_rows=init))
        torch.cuda.synchronize()
        assert torch.equal(f1, f2) and torch.equal(h1, h2)


def test_batched_cluster_chains_equal_single_chains(aoc):
    """launch_cluster_proxies_batch (several frames' k-means advanced as one chain over replicated segments) produces,
    frame by frame, exactly the proxy tables of separate chains."""
    syn, hot = aoc.synthetic, aoc.hotpath
    cfg = syn.CONFIGS["tiny"]
    clip = syn.make_clip(cfg, 8, frames=4)
    emb = torch.from_numpy(clip["emb"]).cuda()
    lab = torch.from_numpy(np.stack([syn.one_hot(l, cfg.n_obj) for l in clip["lab"]])).cuda()
    mc = hot.MatchingConfig()
    side = torch.cuda.Stream()
    R = 2
    inits = []
    for f in range(3):
        rows = syn.kmeans_init_rows(100 + f, [int((clip["lab"][:R] == o).sum()) for o in range(cfg.n_obj)], 16)
        init = np.zeros((cfg.n_obj, 16), np.int32)
        for o, r in enumerate(rows):
            if r is not None:
                init[o, :len(r)] = r
        inits.append(torch.from_numpy(init).cuda())
    batch = hot.launch_cluster_proxies_batch(mc, emb[:R], lab[:R], inits, side)
    singles = [hot.launch_cluster_proxies(mc, emb[:R], lab[:R], i, side) for i in inits]
    torch.cuda.synchronize()
    n = cfg.n_obj * 2 * 16
    for b, s in zip(batch, singles):
        assert torch.equal(b.table[:n], s.table[:n])
        assert torch.equal(b.sqn[:n], s.sqn[:n])


@pytest.mark.parametrize("levels,n_frames", [([16], 3), ([8, 16, 32], 1), ([8, 16, 32], 3), ([32], 7), ([64], 2)])
def test_replica_fused_assignment_is_bit_identical(aoc, levels, n_frames):
    """aoc_kmeans_segmented_rep (the rows of a block fetched once per GROUP of replicas: 6 / 3 / 1 code books per group at K <= 16 / 32 / 64,
    so 7 replicas at K = 32 run as groups of 3 + 3 + 1 and K = 64 falls back to one replica per launch item) against the same replicated
    lists run with n_rep = 1: labels, code books and cluster sizes of all 20 Lloyd iterations' final state are equal bit for bit."""
    syn, ops = aoc.synthetic, aoc.ops
    cfg = syn.CONFIGS["cfg1"]
    clip = syn.make_clip(cfg, 21, frames=11)
    R, O = 3, cfg.n_obj
    idx = [0, 5, 10]
    pool = torch.from_numpy(clip["emb"][idx]).cuda().reshape(-1, cfg.c)
    lab = torch.from_numpy(np.stack([syn.one_hot(clip["lab"][i], O) for i in idx])).cuda().reshape(-1, O)
    prep = ops.label_prep(lab)
    counts = prep.counts.cpu().numpy()
    L, F, kmax = len(levels), n_frames, max(levels)
    cap = prep.obj_rows.numel()
    rows_f, off_f, k_f = ops.kmeans_replicate_levels(prep.obj_rows, prep.obj_offsets, O, F * L, levels, rows_capacity=cap)
    init = np.zeros((F * L * O, kmax), np.int32)
    for f in range(F):
        for li, k in enumerate(levels):
            rows = syn.kmeans_init_rows(1000 + 17 * f + li, [int(c) for c in counts[:O]], k)
            for o, r in enumerate(rows):
                if r is not None:
                    init[(f * L + li) * O + o, :len(r)] = r
    init = torch.from_numpy(init).cuda()
    fused = ops.kmeans_segmented(pool, rows_f, off_f, k_f, init, kmax, 20, rows_capacity=F * L * cap, n_rep=F * L)
    plain = ops.kmeans_segmented(pool, rows_f, off_f, k_f, init, kmax, 20, rows_capacity=F * L * cap, n_rep=1)
    torch.cuda.synchronize()
    n_rows = int(off_f[-1].item())
    assert torch.equal(fused[1][:n_rows], plain[1][:n_rows])
    kk = k_f.cpu().numpy()
    for s in range(F * L * O):
        assert torch.equal(fused[0][s, :kk[s]], plain[0][s, :kk[s]])
        assert torch.equal(fused[2][s, :kk[s]], plain[2][s, :kk[s]])
    # and replicas with the same level and initial rows would agree with each other: replica 0 against a single-replica call
    one = ops.kmeans_replicate_levels(prep.obj_rows, prep.obj_offsets, O, L, levels, rows_capacity=cap)
    single = ops.kmeans_segmented(pool, one[0], one[1], one[2], init[:L * O], kmax, 20, rows_capacity=L * cap, n_rep=1)
    torch.cuda.synchronize()
    for s in range(L * O):
        assert torch.equal(fused[0][s, :kk[s]], single[0][s, :kk[s]])


def test_reused_proxies_accuracy(aoc):
    """NON-PARITY mode: adaptive proxies computed once for a pool and reused for later frames that see the
    same pool, instead of re-clustering with fresh initial rows.  The cluster channels then differ from the reference's by
    construction -- exactly as two runs of the reference with different numpy seeds differ from each other.  The check:
    every other channel is identical, and the reused proxies disagree with a fresh clustering no more than two fresh
    clusterings (different initial rows) disagree with each other (object decision from the nearest-proxy channel, and
    mean absolute feature difference)."""
    syn, hot = aoc.synthetic, aoc.hotpath
    cfg = syn.CONFIGS["cfg1"]
    clip = syn.make_clip(cfg, 5, frames=6)
    emb = torch.from_numpy(clip["emb"]).cuda()
    lab = torch.from_numpy(np.stack([syn.one_hot(l, cfg.n_obj) for l in clip["lab"]])).cuda()
    mc = hot.MatchingConfig()
    bias = torch.zeros(cfg.n_obj, device="cuda")
    side = torch.cuda.Stream()
    ch = hot.channel_slices(mc)
    counts = [int((clip["lab"][0] == o).sum()) for o in range(cfg.n_obj)]

    def init_for(seed):
        rows = syn.kmeans_init_rows(seed, counts, 16)
        init = np.zeros((cfg.n_obj, 16), np.int32)
        for o, r in enumerate(rows):
            if r is not None:
                init[o, :len(r)] = r
        return torch.from_numpy(init).cuda()

    def features(t, handle):
        f, _, _ = hot.proto_mask_features(mc, emb[:1], lab[:1], emb[t - 1], lab[t - 1], emb[t], bias, cluster_ahead=handle)
        torch.cuda.synchronize()
        return f

    cached = hot.launch_cluster_proxies(mc, emb[:1], lab[:1], init_for(1), side)
    c0 = ch["cluster"]
    dec = lambda f: f[:, c0].argmin(0)
    agree_cached, agree_fresh, diff_cached, diff_fresh = [], [], [], []
    for t in range(2, 6):
        f_a = features(t, hot.launch_cluster_proxies(mc, emb[:1], lab[:1], init_for(10 + t), side))
        f_b = features(t, hot.launch_cluster_proxies(mc, emb[:1], lab[:1], init_for(20 + t), side))
        f_c = features(t, cached)
        other = [i for i in range(f_a.shape[1]) if i not in (c0, c0 + 1)]
        assert torch.equal(f_a[:, other], f_c[:, other])
        agree_fresh.append(float((dec(f_a) == dec(f_b)).float().mean()))
        agree_cached.append(float((dec(f_a) == dec(f_c)).float().mean()))
        diff_fresh.append(float((f_a[:, c0:c0 + 2] - f_b[:, c0:c0 + 2]).abs().mean()))
        diff_cached.append(float((f_a[:, c0:c0 + 2] - f_c[:, c0:c0 + 2]).abs().mean()))
    assert np.mean(agree_cached) >= np.mean(agree_fresh) - 0.03, (agree_cached, agree_fresh)
    assert np.mean(diff_cached) <= 1.25 * np.mean(diff_fresh) + 1e-4, (diff_cached, diff_fresh)


def test_pruning_with_ties_duplicates_and_mixed_norms(aoc):
    """The coarse-then-rescore kernel may skip a (reference tile, query tile) pair only when no pixel of it can hold the minimum.
    Adversarial pool: many exact duplicates of the query pixels (ties at distance 0), near-duplicates a few fp16-lo units apart, and
    reference pixels whose norms span two orders of magnitude (the margin is priced with the LARGEST norm).  The raw distances must
    equal the exact-fp32 kernel's within rounding, run after run (the set of rescored tiles depends on timing; the result must not)."""
    rng = np.random.RandomState(11)
    m, c, o = 1500, 100, 3
    q = (np.maximum(rng.randn(m, c), 0) * 0.3).astype(np.float32)
    dup = q[rng.randint(0, m, 6000)]                                           # exact copies of query pixels
    near = dup[:3000] * np.float32(1.0 + 2.0 ** -13) + np.float32(2.0 ** -15)  # differ from a copy only in the lo plane
    big = (np.maximum(rng.randn(4000, c), 0) * 3.0).astype(np.float32)         # norms ~ 10x
    tiny = (np.maximum(rng.randn(3000, c), 0) * 0.003).astype(np.float32)      # norms ~ 1/100
    pool = np.concatenate([dup, near, big, tiny, q[::-1].copy()]).astype(np.float32)
    rng.shuffle(pool)
    ids = rng.randint(0, o, pool.shape[0])
    lab = torch.from_numpy((ids[:, None] == np.arange(o)).astype(np.float32))
    q, pool = torch.from_numpy(q), torch.from_numpy(pool)
    raw_f = _dense(aoc, q, pool, lab, None, "fp32", transform=False)
    first = None
    for _ in range(4):
        raw_s = _dense(aoc, q, pool, lab, None, "split", transform=False)
        assert float((raw_s - raw_f).abs().max()) < 2e-4                       # distances up to ~1e3 here: 2e-4 is a few fp32 ulps of them
        small = raw_f < 1.0                                                     # where the minimum is a (near-)duplicate: plain ATOL_RAW
        assert float((raw_s - raw_f)[small].abs().max()) < ATOL_RAW
        if first is None:
            first = raw_s
        assert torch.equal(raw_s, first)                                        # deterministic although the pruning pattern is not
    st = aoc.ops.dense_prune_stats()
    assert 0 < st["rescored"] <= st["tested"]


@pytest.mark.parametrize("name,R", [("cfg4", 2), ("cfg3", 3)])
def test_full_size_configs_against_exact_fp32(aoc, name, R):
    """BASELINE configs[3] / [2] at full map size (181x321 with 9 objects, 145x261 with 6): the coarse-then-rescore kernel against the
    exact-fp32 kernel on a synthetic clip whose pool is built as the memory policy builds it (every 5th frame)."""
    syn = aoc.synthetic
    cfg = syn.CONFIGS[name]
    clip = syn.make_clip(cfg, 3, frames=(R - 1) * 5 + 3)
    pool = torch.from_numpy(clip["emb"][0:(R - 1) * 5 + 1:5].reshape(-1, cfg.c).copy())
    lab = torch.from_numpy(np.concatenate([syn.one_hot(clip["lab"][i], cfg.n_obj).reshape(-1, cfg.n_obj) for i in range(0, (R - 1) * 5 + 1, 5)]))
    q = torch.from_numpy(clip["emb"][(R - 1) * 5 + 2].reshape(-1, cfg.c).copy())
    bias = torch.linspace(-0.2, 0.2, cfg.n_obj)
    got = _dense(aoc, q, pool, lab, bias, "split")
    want = _dense(aoc, q, pool, lab, bias, "fp32")
    assert got.shape == (cfg.n_obj, cfg.h * cfg.w)
    assert float((got - want).abs().max()) < ATOL


def test_bound_seeds_change_no_result():
    """Round 6: dense_seed_kernel publishes, before the matrix kernel starts, one exact-minus-margin value per query pixel (the same pixel of the newest reference
    frame).  The seeds may only spare work: the development build with AOC_DENSE_SEED=0 and =1 has to produce the same bits for every output -- growing pools,
    a pool whose newest frame IS the query (the seed pair is the unique best, at distance 0), exact duplicates of the best row elsewhere in the pool, a moved
    object (seeds for the wrong object), and a pool that is not whole frames (the seed rule then picks unrelated rows: still real pairs)."""
    import hashlib, os, subprocess, sys
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, %r)
import aoc_amd
from aoc_amd import ops, synthetic as syn
cfg = syn.ClipConfig("t", 45, 77, 4, 16, frames=10)
clip = syn.make_clip(cfg, 5)
emb = torch.from_numpy(clip["emb"]).cuda()
lab = torch.from_numpy(np.stack([syn.one_hot(l, cfg.n_obj) for l in clip["lab"]])).cuda()
hw, C, O = cfg.h * cfg.w, cfg.c, cfg.n_obj
h = hashlib.sha256()
def run(pool, labels, q):
    prep = ops.label_prep(labels.contiguous())
    out = torch.empty(O, q.shape[0], device="cuda")
    ps = ops.split_rows(pool.contiguous())
    qs = ops.split_rows(q.contiguous(), overflow=ps.overflow)
    ops.dense_match_min_split(q.contiguous(), qs, pool.contiguous(), ps, prep, torch.zeros(O, device="cuda"), out, 1, q.shape[0], True)
    torch.cuda.synchronize()
    h.update(out.cpu().numpy().tobytes())
for R in (1, 2, 4):                                             # growing pools, query three frames after the newest pool frame
    run(emb[:R * 2:2].reshape(-1, C), lab[:R * 2:2].reshape(-1, O), emb[R * 2 + 1].reshape(-1, C))
run(emb[[0, 3]].reshape(-1, C), lab[[0, 3]].reshape(-1, O), emb[3].reshape(-1, C))              # the newest pool frame is the query itself
pool = emb[[0, 3]].reshape(-1, C).clone(); pool[100:140] = pool[hw + 100:hw + 140]                # duplicates of best rows in the older frame
run(pool, lab[[0, 3]].reshape(-1, O), emb[3].reshape(-1, C))
moved = torch.roll(lab[3], shifts=(7, 11), dims=(0, 1))                                           # labels moved: seeds land on other objects
run(emb[[0, 3]].reshape(-1, C), torch.stack([lab[0], moved]).reshape(-1, O), emb[4].reshape(-1, C))
run(emb[[0, 3]].reshape(-1, C)[: 2 * hw - 333], lab[[0, 3]].reshape(-1, O)[: 2 * hw - 333], emb[4].reshape(-1, C))   # not whole frames
st = ops.dense_prune_stats()
print("HASH", h.hexdigest(), st["rescored"], st["tested"])
''' % root
    res = {}
    for seed in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, AOC_LIB_VARIANT="dev", AOC_DENSE_SEED=seed), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
        line = [l for l in r.stdout.splitlines() if l.startswith("HASH")][-1].split()
        res[seed] = (line[1], int(line[2]), int(line[3]))
    assert res["0"][0] == res["1"][0], "the seeds changed a result"
    assert res["0"][2] == res["1"][2] and res["1"][1] < res["0"][1], res          # same pairs tested, fewer rescored
