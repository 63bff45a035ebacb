"""aoc_proxy_corr_min_batched (fp16-split matrix pipe, several frames per launch) against the CPU oracle's distances and against the
exact-fp32 kernel.  Tolerance on the proto-mask outputs: 5e-6, as for every other matching branch."""
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


def _oracle(q, table, sqn, sb, ss, bias, transform=True):
    """[n_set, m] float64-free restatement with the oracle's own functions (AEM:29-59, 92-128)."""
    from oracle import matching as om
    qsq = q.pow(2).sum(1)
    outs = []
    for s, (b, n) in enumerate(zip(sb, ss)):
        live = [i for i in range(b, b + n) if np.isfinite(float(sqn[i]))]
        if not live:
            d = torch.full((q.shape[0],), 5e4)
        else:
            p = table[live]
            d = om.flattened_pairwise_distances(p, p.pow(2).sum(1), q, qsq).min(dim=1)[0]
        outs.append(om.proto_transform(d, bias[s]) if transform else d)
    return torch.stack(outs)


def _case(rng, m, C, sizes, absent=(), n_frames=1, scale=0.3):
    """sets of the given sizes over one proxy table (kmax-strided like the product's tables), `absent` proxies get norm = +inf"""
    kmax = max(max(sizes), 1)
    sb = [i * kmax for i in range(len(sizes))]
    n_proxy = len(sizes) * kmax
    frames = []
    for _ in range(n_frames):
        q = torch.from_numpy((np.maximum(rng.randn(m, C), 0) * scale).astype(np.float32))
        t = torch.from_numpy((np.maximum(rng.randn(n_proxy, C), 0) * scale).astype(np.float32))
        sq = t.pow(2).sum(1)
        for a in absent:
            sq[a] = float("inf")
        bias = torch.from_numpy((rng.rand(len(sizes)).astype(np.float32) - 0.5))
        frames.append((q, t, sq, bias))
    return frames, sb, list(sizes)


def _run_split(aoc, entry, dev_frames, sb, ss, so, transform=True):
    """the fp16-split correlation through either entry point: fp32 queries (aoc_proxy_corr_min_batched) or the queries as tile-major
    split records (aoc_proxy_corr_min_records)"""
    if entry == "records":
        rec = [(f[0], aoc.ops.split_rows(f[0], tiled=True), f[1], f[2], f[3], f[4]) for f in dev_frames]
        aoc.ops.proxy_corr_min_records(rec, sb, ss, so, transform)
    else:
        aoc.ops.proxy_corr_min_batched(dev_frames, sb, ss, so, transform, "split")


@pytest.mark.parametrize("entry", ["batched", "records"])
@pytest.mark.parametrize("sizes,absent,m,n_frames", [
    ([16] * 8 + [1] * 4, (), 25773, 2),                  # cfg2 shape: 4 objects x (centroid, centroid_avg) + 4 k = 1 proxies
    ([8] * 6 + [16] * 6 + [32] * 6 + [1] * 3, (), 1000, 3),   # multi-level
    ([64, 64, 5, 12, 1, 1, 20, 40, 0, 3], (), 777, 2),   # multi-tile sets, odd sizes, an empty set
    ([16, 16, 16, 16, 1, 1], (3, 4, 5, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 65), 333, 1),   # absent proxies / whole absent set / absent k=1
    ([16] * 8 + [1] * 4, (), 31, 5),                     # fewer pixels than a tile
])
def test_batched_split_vs_oracle(aoc, sizes, absent, m, n_frames, entry):
    rng = np.random.RandomState(len(sizes) * 7 + m)
    C = 100
    frames, sb, ss = _case(rng, m, C, sizes, absent, n_frames)
    n_set = len(sizes)
    so = [s * m for s in range(n_set)]
    dev_frames, outs = [], []
    for q, t, sq, bias in frames:
        out = torch.full((n_set, m), -7.0, device="cuda")
        outs.append(out)
        dev_frames.append((q.cuda(), t.cuda(), sq.cuda(), bias.cuda(), out))
    _run_split(aoc, entry, dev_frames, sb, ss, so)
    ref32 = [torch.full((n_set, m), -7.0, device="cuda") for _ in frames]
    aoc.ops.proxy_corr_min_batched([(f[0], f[1], f[2], f[3], r) for f, r in zip(dev_frames, ref32)], sb, ss, so, True, "fp32")
    for (q, t, sq, bias), out, r32 in zip(frames, outs, ref32):
        want = _oracle(q, t, sq, sb, ss, bias)
        np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=0, atol=ATOL)
        np.testing.assert_allclose(r32.cpu().numpy(), want.numpy(), rtol=0, atol=ATOL)


@pytest.mark.parametrize("entry", ["batched", "records"])
def test_batched_raw_distances_and_wide_range(aoc, entry):
    """transform = 0 (raw squared distances) on data with a wide dynamic range, against float64."""
    rng = np.random.RandomState(5)
    m, C = 2000, 100
    q = (rng.randn(m, C) * np.exp(rng.randn(m, 1))).astype(np.float32) * 0.5
    t = (rng.randn(40, C) * np.exp(rng.randn(40, 1))).astype(np.float32) * 0.5
    sb, ss = [0, 16, 32, 33, 34], [16, 16, 1, 1, 6]
    so = [s * m for s in range(5)]
    out = torch.empty(5, m, device="cuda")
    _run_split(aoc, entry, [(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda(), None, None, out)], sb, ss, so, False)
    q64, t64 = q.astype(np.float64), t.astype(np.float64)
    d = (q64 ** 2).sum(1)[:, None] + (t64 ** 2).sum(1)[None] - 2 * q64 @ t64.T
    want = np.stack([d[:, b:b + n].min(1) for b, n in zip(sb, ss)])
    scale = (q64 ** 2).sum(1)[None] + (t64 ** 2).sum(1).max()
    assert float(np.max(np.abs(out.cpu().numpy() - want) / scale)) < 2e-6      # fp32-level relative error of a distance


@pytest.mark.parametrize("entry", ["batched", "records"])
def test_batched_takeover_when_values_do_not_fit(aoc, entry):
    """|x| * 2^10 > 65000 somewhere: the device-side flag makes the exact-fp32 kernel recompute the launch (no host round trip)."""
    rng = np.random.RandomState(6)
    m, C = 500, 100
    frames, sb, ss = _case(rng, m, C, [16, 16, 1], (), 2)
    frames[1][0][123, 7] = 80.0                                   # 80 * 1024 > 65000
    so = [s * m for s in range(3)]
    dev_frames = [(q.cuda(), t.cuda(), sq.cuda(), b.cuda(), torch.empty(3, m, device="cuda")) for q, t, sq, b in frames]
    _run_split(aoc, entry, dev_frames, sb, ss, so)
    ref = [torch.empty(3, m, device="cuda") for _ in frames]
    aoc.ops.proxy_corr_min_batched([(f[0], f[1], f[2], f[3], r) for f, r in zip(dev_frames, ref)], sb, ss, so, True, "fp32")
    for f, r in zip(dev_frames, ref):
        assert torch.equal(f[4], r)                               # bit-identical: it IS the fp32 kernel's result


@pytest.mark.gpu
def test_sets_beyond_the_lds_image_take_the_exact_kernel(aoc):
    """A set with more than 160 proxies does not fit the split kernel's LDS image: the whole call must run on the exact-fp32 kernel
    (IncrementalProxyBank builds sets of R x K proxies, 192 at R = 12) and still match the oracle."""
    import torch
    from oracle import matching as om
    torch.manual_seed(3)
    m, C = 1500, 100
    q = torch.relu(torch.randn(m, C)) * 0.3
    tab = torch.relu(torch.randn(200 + 16 + 1, C)) * 0.3
    sb, ss = [0, 200, 216], [200, 16, 1]
    bias = torch.tensor([0.1, -0.2, 0.0])
    out = torch.empty(3, m).cuda()
    aoc.ops.proxy_corr_min_batched([(q.cuda(), tab.cuda(), None, bias.cuda(), out)], sb, ss, [s * m for s in range(3)])
    dist = om.flattened_pairwise_distances(tab, tab.pow(2).sum(1), q, q.pow(2).sum(1))
    want = torch.stack([om.proto_transform(dist[:, b:b + n].min(1)[0], bias[i]) for i, (b, n) in enumerate(zip(sb, ss))])
    assert torch.allclose(out.cpu(), want, rtol=0, atol=5e-6)


def test_tiled_records_are_the_row_major_records_reordered(aoc):
    """aoc_split_rows_tiled writes exactly the chunks of aoc_split_rows, tile-major ([tile][plane][k-step][k-half][row % 32]); rows past n
    are zero records; sqnorm and the overflow flag agree."""
    rng = np.random.RandomState(11)
    n, C = 1000 + 7, 100
    x = torch.from_numpy((rng.randn(n, C) * 0.4).astype(np.float32)).cuda()
    a = aoc.ops.split_rows(x)
    b = aoc.ops.split_rows(x, tiled=True)
    assert torch.equal(a.sqnorm, b.sqnorm) and int(b.overflow.item()) == 0
    T = (n + 31) // 32
    rows = torch.zeros(T * 32, 448, dtype=torch.uint8, device="cuda")
    rows[:n] = a.records
    # row-major chunk (plane p, k-step s, half h) of row r = bytes [(p * 14 + s * 2 + h) * 16, +16)
    want = rows.view(T, 32, 2, 7, 2, 16).permute(0, 2, 3, 4, 1, 5).contiguous().view(T * 32, 448)
    assert torch.equal(b.records, want)


@pytest.mark.parametrize("levels,n_obj,kmax", [([8, 16, 32], 6, 32), ([64], 9, 64), ([16], 4, 16), ([8, 16, 32], 2, 32), ([64], 24, 64)])   # the last: 97 tiles = more passes than the table holds
def test_all_passes_in_one_launch_equal_pass_by_pass(levels, n_obj, kmax):
    """aoc_proxy_corr_min_records_cached (round 5: the passes of a frame with more than 5 proxy tiles as a grid dimension, tile tables kept in the
    workspace) against aoc_proxy_corr_min_records (one launch per pass): every workgroup runs the same code on the same pass -> EQUAL, on the
    first call (tables written), on a second call with other data (tables reused) and after the set structure changes (tables rewritten)."""
    import aoc_amd
    ops = aoc_amd.ops
    torch.manual_seed(3)
    h, w, C = 37, 53, 100
    m = h * w
    L = len(levels)
    n_ad = L * n_obj * 2 * kmax
    sb, ss, so = [], [], []
    n_ch = 2 * L + 1
    for l, k in enumerate(levels):
        for o in range(n_obj):
            for f in range(2):
                sb.append(((l * n_obj + o) * 2 + f) * kmax)
                ss.append(k)
                so.append((o * n_ch + 2 * l + f) * m)
    for o in range(n_obj):
        sb.append(n_ad + o)
        ss.append(1)
        so.append((o * n_ch + 2 * L) * m)
    cache = ops.CorrTableCache(torch.device("cuda"))
    for trial in range(3):
        q = (torch.rand(m, C, device="cuda") * 0.3).contiguous()
        table = (torch.rand(n_ad + n_obj, C, device="cuda") * 0.3).contiguous()
        sqn = table.pow(2).sum(1)
        sqn[torch.rand_like(sqn) < 0.1] = float("inf")               # absent proxies
        bias = torch.randn(len(sb), device="cuda") * 0.2
        qs = ops.split_rows(q, tiled=True)
        sets = (sb, ss, so) if trial < 2 else (sb[:-1], ss[:-1], so[:-1])       # third trial: another set structure -> tables rewritten
        b = bias[:len(sets[0])].contiguous()
        out_a = torch.zeros(n_obj, n_ch, m, device="cuda")
        out_b = torch.zeros_like(out_a)
        ops.proxy_corr_min_records([(q, qs, table, sqn, b, out_a)], *sets, True)
        key_before = cache.key.value
        ops.proxy_corr_min_records([(q, qs, table, sqn, b, out_b)], *sets, True, cache=cache)
        torch.cuda.synchronize()
        assert torch.equal(out_a, out_b), f"trial {trial}: one launch != pass by pass ({float((out_a - out_b).abs().max())})"
        assert cache.key.value != 0
        if n_obj < 24:            # (beyond 16 passes the tables go out in two launches and the key is the last one's)
            assert (trial != 1 or cache.key.value == key_before) and (trial != 2 or cache.key.value != key_before)
