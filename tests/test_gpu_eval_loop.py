"""Eval-loop memory policy on the GPU against its CPU restatement (oracle/eval_loop.py)."""
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


def _probs(rng, n_ch, H, W, sharp=3.0):
    logits = torch.from_numpy(rng.randn(n_ch, H, W).astype(np.float32)) * sharp
    return torch.softmax(logits, dim=0)


@pytest.mark.parametrize("n_ch,seen,with_join", [(5, [0, 1, 2, 3, 4], False), (6, [0, 2, 3], False), (4, [0, 1], True), (11, list(range(8)), True)])
def test_confident_labels_vs_oracle(aoc, n_ch, seen, with_join):
    from oracle import eval_loop as oe
    rng = np.random.RandomState(n_ch)
    H, W = 37, 53
    probs = _probs(rng, n_ch, H, W)
    probs[:, 0, :5] = 1.0 / n_ch                      # exact ties: first maximum wins
    probs[:, 1, :5] = 0.0                             # all-zero pixel
    join = None
    if with_join:
        j = np.zeros((H, W), np.int64)
        j[5:12, 7:20] = n_ch - 1                      # a new object appears
        j[20:24, 30:40] = -1                          # "unsure" region of the annotation
        join = torch.from_numpy(j)
    bits = sum(1 << s for s in seen)
    for unc in (0.3, 1.0):
        lab, conf, ent = aoc.ops.confident_labels(probs.reshape(n_ch, -1).cuda(), bits, None if join is None else join.cuda(), unc)
        wl, wc, wu = oe.frame_decision(probs[None], seen, join, unc)
        assert np.array_equal(lab.cpu().numpy().reshape(H, W), wl.numpy())
        np.testing.assert_allclose(ent.cpu().numpy().reshape(H, W), wu.numpy(), rtol=0, atol=2e-6)
        # 125 marks can only differ where the entropy sits within rounding of the threshold
        differ = conf.cpu().numpy().reshape(H, W) != wc.numpy()
        assert not differ.any() or float(np.abs(wu.numpy()[differ] - unc).max()) < 2e-6


@pytest.mark.parametrize("H,W,h,w,n_obj", [(480, 854, 121, 214, 4), (97, 131, 25, 33, 7), (33, 33, 33, 33, 3)])
def test_label_onehot_nearest_vs_oracle(aoc, H, W, h, w, n_obj):
    from oracle import eval_loop as oe
    rng = np.random.RandomState(H)
    lab = torch.from_numpy(rng.randint(0, n_obj, (H, W)).astype(np.int32))
    lab[rng.rand(H, W) < 0.05] = 125                  # uncertain pixels match no object
    got = aoc.ops.label_onehot_nearest(lab.cuda(), h, w, n_obj).cpu()
    want = oe.label_onehot_nearest(lab, h, w, n_obj)
    assert torch.equal(got, want)


def test_memory_policy_sequence_vs_oracle(aoc):
    """A 12-frame synthetic sequence with a new object joining at frame 4: pool growth, confident maps and the tensors
    handed to the matching path equal the restatement; the pool then runs through proto_mask_features."""
    from oracle import eval_loop as oe
    syn, hot = aoc.synthetic, aoc.hotpath
    cfg = syn.CONFIGS["tiny"]
    clip = syn.make_clip(cfg, 9, frames=12)
    rng = np.random.RandomState(0)
    H, W = cfg.h * 4, cfg.w * 4
    n_ch = cfg.n_obj
    def upsample(ids):
        return torch.from_numpy(np.kron(ids, np.ones((4, 4), np.int64))[:H, :W].astype(np.int32))
    gpu = aoc.eval_loop.MemoryPolicy(mem_every=3, unc_ratio=0.6)
    cpu = oe.MemoryPolicy(mem_every=3, unc_ratio=0.6)
    emb = [torch.from_numpy(e) for e in clip["emb"]]
    first = upsample(clip["lab"][0])
    first[first == n_ch - 1] = 0                      # the last object is not annotated in frame 0
    gpu.start(emb[0].cuda(), first.cuda())
    cpu.start(emb[0], first.long())
    for t in range(1, 12):
        onehot = torch.from_numpy(syn.one_hot(clip["lab"][t], n_ch)).permute(2, 0, 1)
        logits = torch.nn.functional.interpolate(onehot[None] * 4.0, size=(H, W), mode="bilinear", align_corners=True)[0]
        probs = torch.softmax(logits + torch.from_numpy(rng.randn(n_ch, H, W).astype(np.float32)) * 0.5, dim=0)
        gt = None
        if t == 4:
            g = upsample(clip["lab"][t])
            g[g != n_ch - 1] = 0                      # ground truth that only introduces the new object
            gt = g
        gl, gc, _ = gpu.update(emb[t].cuda(), probs.cuda(), None if gt is None else gt.cuda())
        cl, cc, _ = cpu.update(emb[t], probs, None if gt is None else gt.long())
        assert np.array_equal(gl.cpu().numpy(), cl.numpy())
        assert float((gc.cpu() != cc).float().mean()) < 1e-4      # entropy within rounding of the threshold
    assert len(gpu.ref_embeddings) == len(cpu.ref_embeddings) == 1 + 1 + 3      # frame 0, GT frame 4, frames 3, 6, 9
    assert sorted(gpu.label_all) == sorted(cpu.label_all_list)
    ref_emb, ref_lab, prev_emb, prev_lab = gpu.reference_pool(cfg.h, cfg.w, n_ch)
    assert tuple(ref_lab.shape) == (5, cfg.h, cfg.w, n_ch)
    want0 = oe.label_onehot_nearest(cpu.ref_mask_confident[0], cfg.h, cfg.w, n_ch)
    assert torch.equal(ref_lab[0].cpu(), want0)
    feat, head, _ = hot.proto_mask_features(hot.MatchingConfig(), ref_emb, ref_lab, prev_emb, prev_lab, emb[11].cuda(),
                                            torch.zeros(n_ch, device="cuda"), init_rows=None)
    assert bool(torch.isfinite(feat).all()) and tuple(feat.shape) == (n_ch, 24, cfg.h, cfg.w)


def test_entropy_kernel_vs_reference_golden(aoc):
    """aoc_confident_labels' entropy map against the output of the reference's cal_shannon_entropy (committed golden vector)."""
    from conftest import load_golden
    g = load_golden("shannon_entropy")
    p = torch.from_numpy(g["preds"])[0]
    n_ch, H, W = p.shape
    _, _, ent = aoc.ops.confident_labels(p.reshape(n_ch, -1).cuda(), (1 << n_ch) - 1, None, 0.5)
    np.testing.assert_allclose(ent.cpu().numpy().reshape(H, W), g["uncertainty"][0, 0], rtol=0, atol=2e-6)


@pytest.mark.parametrize("mode,after_relu", [("l2", False), ("l1", False), ("l1", True)])
def test_gct_vs_oracle(aoc, mode, after_relu):
    from oracle import calibration as ocal
    rng = np.random.RandomState(3)
    N, C, H, W = 3, 48, 31, 45
    x = torch.from_numpy(rng.randn(N, C, H, W).astype(np.float32))
    if after_relu:
        x = x.clamp_min(0)
    m = aoc.gct.GCT(C, mode=mode, after_relu=after_relu).cuda()
    with torch.no_grad():
        m.alpha.copy_(torch.from_numpy(rng.rand(1, C, 1, 1).astype(np.float32)) + 0.5)
        m.gamma.copy_(torch.from_numpy(rng.randn(1, C, 1, 1).astype(np.float32)))
        m.beta.copy_(torch.from_numpy(rng.randn(1, C, 1, 1).astype(np.float32)) * 0.3)
        got = m(x.cuda()).cpu()
        want = ocal.gct_forward(x, m.alpha.cpu(), m.gamma.cpu(), m.beta.cpu(), m.epsilon, mode, after_relu)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-5, atol=2e-6)


def test_ia_logit_vs_oracle(aoc):
    from oracle import calibration as ocal
    rng = np.random.RandomState(4)
    N, C, H, W, D = 4, 37, 29, 41, 400
    x = torch.from_numpy(rng.randn(N, C, H, W).astype(np.float32))
    head = torch.from_numpy(rng.randn(N, D).astype(np.float32))
    lin = torch.nn.Linear(D, C + 1)
    with torch.no_grad():
        got = aoc.gct.IA_logit(x.cuda(), head.cuda(), lin.cuda()).cpu()
        lin = lin.cpu()
        want = ocal.ia_logit(x, head, lin.weight, lin.bias)
    assert tuple(got.shape) == (N, 1, H, W)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("O,h,w", [(4, 121, 213), (3, 17, 23)])
def test_dynamic_prehead_vs_torch(aoc, O, h, w):
    """hotpath.DynamicPreHead against the same torch modules the reference class is made of (decoding_module.py:228-240),
    and the fused concatenation of aocnet.py:362."""
    rng = np.random.RandomState(O)
    x = torch.from_numpy(rng.uniform(-1, 1, (O, 24, h, w)).astype(np.float32))
    emb = torch.from_numpy((np.maximum(rng.randn(h, w, 100), 0) * 0.3).astype(np.float32))
    torch.manual_seed(0)
    m = aoc.hotpath.DynamicPreHead(in_dim=24, embed_dim=64)
    with torch.no_grad():
        m.bn.weight.copy_(torch.from_numpy(rng.rand(64).astype(np.float32)) + 0.5)
        m.bn.bias.copy_(torch.from_numpy(rng.randn(64).astype(np.float32)) * 0.2)
        want = torch.relu(m.bn(m.conv(x)))                                   # the reference forward, on the CPU
        mg = m.cuda()
        got = mg(x.cuda()).cpu()
        cat = mg(x.cuda(), emb.cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4, atol=2e-5)
    assert tuple(cat.shape) == (O, 164, h, w)
    assert torch.equal(cat[:, 100:], got)
    assert torch.equal(cat[:, :100], emb.permute(2, 0, 1).unsqueeze(0).expand(O, -1, -1, -1))


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """The multi-rank path of bench.py (rank set-up, barriers, max-over-ranks timing, all-reduce of the counters, rank-0 report) with two
    ranks sharing the one GPU of the test box: AOC_DIST_BACKEND=gloo is the developer switch for exactly this; RCCL needs one GPU per rank."""
    import json, os, subprocess, sys
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AOC_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--exact-steps", "0", "--no-extras",
                        "--details-file", "gpurun_out/bench_details_gloo2.json"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["scaling"] == "weak"
    assert line["value"] > 0 and abs(line["value"] - 2 * line["config"]["sequences_per_gpu"] * 1e3 / line["ms_per_step"]) < 0.05 * line["value"]
    assert line["ranks_seen"] == 2 and line["frames_per_rank"] == [4 * line["config"]["sequences_per_gpu"]] * 2 and line["imbalance"] >= 1.0
    assert len(r.stdout.splitlines()[-1]) < 4096 and line["cpu_baseline"] is None and os.path.exists(os.path.join(root, line["details_file"]))


def _run_py(code, env=None, timeout=900):
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=timeout, cwd=root)


def test_metric_allreduce_through_rccl_with_one_rank():
    """The only collectives of the sequence-sharded evaluation (one all-reduce(SUM) of the metric accumulators + one all-reduce(MAX) of the
    rank seconds, eval_manager_mm.py:172 is the shard point) through RCCL itself -- backend "nccl", device tensors -- with the one rank a one-GPU box
    can host, then the whole sharded runner on top of it: RCCL plumbing (device_id binding, float64 reductions, barrier, teardown) has run before the
    first multi-GPU node does."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    code = r'''
import os, socket, sys
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
import aoc_amd
from aoc_amd import sharding, eval_runner
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
m = sharding.allreduce_metrics(dict(frames=7, objects=21, gpu_seconds=0.5, sum_iou=3.25, iou_count=4, sum_f=1.5), device=dev)
assert m == dict(frames=7.0, objects=21.0, gpu_seconds=0.5, sum_iou=3.25, iou_count=4.0, sum_f=1.5), m
assert sharding.allreduce_max(2.75, device=dev) == 2.75
specs = eval_runner.make_sequence_set("davis17", scale=0.07, seed=0)
def barrier():
    torch.cuda.synchronize(); dist.barrier()
with torch.no_grad():
    tot = eval_runner.eval_sharded(specs, 0, 1, dev, barrier=barrier, lanes=2, max_frames=6)
assert tot["ranks"] == 1 and tot["frames"] > 0 and 0.0 <= tot["mean_j"] <= 1.0, tot
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK", tot["frames"], tot["mean_j"])
'''
    r = _run_py(code, env=dict(HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]


def test_bench_line_with_the_process_group_on_rccl():
    """bench.py's multi-rank code path (process-group set-up with device_id, barriers, the region-count agreement, max-over-ranks timing, the metric and
    per-rank all-reduces, rank-0 report, teardown) with backend "nccl" = RCCL, forced on for the single rank of a one-GPU box (AOC_DIST_FORCE=1)."""
    import json, os, subprocess, sys
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AOC_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("AOC_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--exact-steps", "0", "--no-extras",
                        "--details-file", "gpurun_out/bench_details_rccl1.json"], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    line = json.loads(r.stdout.strip().splitlines()[-1])            # the report is the LAST line of stdout, after whatever RCCL prints about itself
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["frames_per_rank"] == [4 * line["config"]["sequences_per_gpu"]] and line["imbalance"] == 1.0
    assert line["value"] > 0 and line["roofline"]["bound"] == "mfma" and line["roofline_correlation"]["bound"] == "hbm"


@pytest.mark.gpu
@pytest.mark.parametrize("levels,n_obj", [((16,), 3), ((8, 16, 32), 4)])
def test_chains_ahead_equal_the_per_frame_reference_api_path(levels, n_obj):
    """HotPathBackend(ahead=True) reads the row counts once per pool state, draws the initial rows of the frames that will see that pool
    in the reference's order and enqueues their k-means chains ahead on a side stream, and every frame is ONE aoc_frame_enqueue call; the
    predicted label maps must equal those of the per-frame path (label prep + read-back + chain on every frame, Python-orchestrated
    individual entry points) bit for bit -- single- and multi-level proxies."""
    import torch
    from aoc_amd import eval_runner as er
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    dev = torch.device("cuda", 0)
    spec = er.SequenceSpec("davis-like", 41, 57, n_obj, 14, seed=5, levels=levels, mem_every=5)
    data = er.load_sequence(spec, dev)
    outs = []
    for ahead in (False, True):
        be = er.HotPathBackend(dev, ahead=ahead)
        be.start(spec)
        be.first_frame(data[0][0], data[1][0])
        outs.append([be.frame(data[0][t]).clone() for t in range(1, spec.frames)])
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert torch.equal(a, b)
