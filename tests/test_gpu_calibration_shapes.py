"""Round 4: the streaming kernels of the calibration path at the plane sizes where their code paths switch -- film_scale_ahead_kernel<4|8>
(one to several workgroups per plane, every 4-byte phase of a 16-byte line, planes shorter than one float4, an output whose alignment
differs from the input's), plane_mean4_kernel, and the k-th-largest selection of the conditioning gate (cond_select_tail_kernel<8|32|64>
and the one-launch-per-digit path behind it) with ties, k = 1 and k = HW.  Checked against plain torch on the same inputs (CL:23-43,
ATT:12-17); the selection is exact, the streams are elementwise products of identical float32 factors."""
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


def _film_ref(x, head, w, b):
    g = 1.0 + torch.tanh(head.double() @ w.double().t() + (b.double() if b is not None else 0.0))          # [O, c]
    return g.float()


def _check_film(x, y, want_gain):
    """Every plane of y is ONE float32 gain times the plane of x, exactly (the gain is the kernel's own float32 dot product: recovered from
    the first element, it must reproduce the whole plane bit for bit), and that gain is the reference's within float32 rounding."""
    g0 = (y[:, :, 0].double() / x[:, :, 0].double()).float()
    ok = torch.zeros(g0.shape, dtype=torch.bool, device=x.device)
    for cand in (g0, torch.nextafter(g0, torch.full_like(g0, float("inf"))), torch.nextafter(g0, torch.full_like(g0, float("-inf")))):
        hit = (y == cand[:, :, None] * x).all(-1)
        g0 = torch.where(hit & ~ok, cand, g0)
        ok |= hit
    assert bool(ok.all()), "a plane is not one gain times its input"
    np.testing.assert_allclose(g0.cpu().numpy(), want_gain.cpu().numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("hw", [1, 3, 4, 5, 63, 1023, 4099, 8191, 8192, 8193, 8197, 12289, 25773, 40003])
@pytest.mark.parametrize("D", [1, 400, 1030])
def test_film_scale_plane_sizes(aoc, hw, D):
    """y = (1 + tanh(head . W^T + b)) x for plane sizes around hw / 4 = 2048 (U = 8 -> 4), odd sizes (planes at every phase of a 16-byte
    line) and planes shorter than a float4; D > 1024 takes the dot product's second trip."""
    torch.manual_seed(hw * 7 + D)
    O, c = 3, 5
    x = torch.randn(O, c, hw, device="cuda")
    head = torch.randn(O, D, device="cuda") * 0.2
    w = torch.randn(c, D, device="cuda") * 0.2
    b = torch.randn(c, device="cuda") * 0.1
    y = aoc.ops.film_scale(x.view(O, c, hw, 1), head, w, b).view(O, c, hw)
    _check_film(x, y, _film_ref(x, head, w, b))


def test_film_scale_output_alignment_differs(aoc):
    """Input planes and output planes at different phases of a 16-byte line: the scalar path."""
    torch.manual_seed(3)
    O, c, hw, D = 2, 3, 1001, 40
    xbuf = torch.randn(O * c * hw + 8, device="cuda")
    ybuf = torch.zeros(O * c * hw + 8, device="cuda")
    head, w, b = torch.randn(O, D, device="cuda"), torch.randn(c, D, device="cuda") * 0.1, torch.zeros(c, device="cuda")
    for xo, yo in ((0, 1), (1, 0), (2, 3), (3, 3)):
        x = xbuf[xo:xo + O * c * hw].view(O, c, hw, 1)
        y = ybuf[yo:yo + O * c * hw].view(O, c, hw, 1)
        out = aoc.ops.film_scale(x, head, w, b, out=y)
        assert out.data_ptr() == y.data_ptr()
        _check_film(x.view(O, c, hw), y.view(O, c, hw), _film_ref(x, head, w, b))


@pytest.mark.parametrize("hw", [1, 2, 5, 255, 2049, 8193, 25773])
def test_plane_mean_sizes(aoc, hw):
    torch.manual_seed(hw)
    x = torch.randn(3, 7, hw, 1, device="cuda")
    got = aoc.ops.plane_mean(x)
    want = x.double().mean(dim=(2, 3))
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("hw", [7, 100, 8191, 8192, 8193, 32768, 32769, 65536, 65537, 70001])
@pytest.mark.parametrize("mode", ["random", "ties", "binade"])
def test_cond_threshold_is_the_kth_largest(aoc, hw, mode):
    """threshold = topk(scores, k)[..., -1] exactly (CL:33) for every kernel of the selection: keys in registers at 8 / 32 / 64 per thread
    (HW <= 8192 / 32768 / 65536), one launch per digit above; with many equal scores (the k-th largest inside a run of ties), scores that
    share their top byte, k = 1 and k = HW."""
    torch.manual_seed(hw + len(mode))
    N, C = 2, 4
    z = torch.randn(N, C, hw, 1, device="cuda")
    if mode == "ties":
        z = torch.round(z * 2) / 2                      # few distinct values: long runs of equal scores
    elif mode == "binade":
        z = 1.0 + torch.rand(N, C, hw, 1, device="cuda") * 0.25      # all scores in one binade: the top byte decides nothing
    phi_w = torch.tensor([1.0, 0.5, -0.25, 2.0], device="cuda") if mode != "binade" else torch.tensor([0.25, 0.25, 0.25, 0.25], device="cuda")
    phi_b = torch.tensor([0.125], device="cuda")
    for k in sorted({1, 2, max(1, int(0.3 * hw)), hw - 1 if hw > 1 else 1, hw}):
        gap, scores, thr = aoc.ops.cond_gate_pool(z, phi_w, phi_b, k, want_debug=True)
        s = scores.cpu()
        want_thr = torch.topk(s, k, dim=1).values[:, -1]
        assert torch.equal(thr.cpu(), want_thr), (hw, mode, k)
        mask = (s > want_thr[:, None]).double()
        want_gap = (z.view(N, C, hw).cpu().double() * mask[:, None, :]).sum(-1) / hw
        np.testing.assert_allclose(gap.cpu().numpy(), want_gap.numpy(), rtol=2e-5, atol=2e-6)


def test_cond_gate_pool_back_to_back_calls_share_a_workspace(aoc):
    """The histograms are zeroed by the first launch of a call (no memset): two calls on different inputs through the same cached workspace
    give what each gives alone."""
    torch.manual_seed(11)
    N, C, hw = 3, 8, 5000
    za, zb = torch.randn(N, C, hw, 1, device="cuda"), torch.randn(N, C, hw, 1, device="cuda") * 3 + 1
    pw, pb = torch.randn(C, device="cuda"), torch.zeros(1, device="cuda")
    k = 1500
    alone = [aoc.ops.cond_gate_pool(z, pw, pb, k, want_debug=True) for z in (za, zb)]
    torch.cuda.synchronize()
    for _ in range(3):
        for z, (gap, scores, thr) in zip((za, zb), alone):
            g2, s2, t2 = aoc.ops.cond_gate_pool(z, pw, pb, k, want_debug=True)
            assert torch.equal(g2, gap) and torch.equal(s2, scores) and torch.equal(t2, thr)
