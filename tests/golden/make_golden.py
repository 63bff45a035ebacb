#!/usr/bin/env python3
"""Generate the committed golden vectors by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the GPU box never sees the
reference, only the ``*.npz`` files this script writes next to itself.  The reference files
are imported by path and executed unmodified; nothing of their text is stored.

    python tests/golden/make_golden.py

What is recorded per case: the exact inputs (float16-representable values stored as float16,
integer label maps as int16), the reference function's output, and -- for the k-means path --
every ``kmeans2`` call the reference made (observed through a recording wrapper that forwards
to the real scipy function): n_i, K_i, the initial rows scipy drew, final labels and code book,
and the labels scipy produces after 1 and 2 iterations from the same initial rows.
"""
import importlib.util
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/AOC-Net"

import aoc_amd  # noqa: E402
from aoc_amd import synthetic as syn  # noqa: E402

warnings.simplefilter("ignore")
torch.set_num_threads(4)


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


aem = load("ref_aem", f"{REF}/adaptive_embedding_for_matching.py")
mt = load("ref_mt", f"{REF}/complete_project/AOCNet/networks/layers/matching.py")
att = load("ref_att", f"{REF}/complete_project/AOCNet/networks/layers/attention.py")
clb = load("ref_clb", f"{REF}/complete_project/AOCNet/networks/aoc/conditioning_layer.py")

import scipy.cluster.vq as _vq  # noqa: E402

_real_kmeans2 = _vq.kmeans2
KM_LOG = []


def recording_kmeans2(data, k, iter=10, minit='random', **kw):
    """Forwards to scipy's kmeans2 unchanged; additionally works out which rows 'points' drew
    (from a copy of the RNG state) and the labels after 1 and 2 iterations."""
    state = np.random.get_state()
    centroid, label = _real_kmeans2(data, k, iter=iter, minit=minit, **kw)
    after = np.random.get_state()
    np.random.set_state(state)
    rows = np.random.permutation(data.shape[0])[:int(k)]
    np.random.set_state(after)
    c2, l2 = _real_kmeans2(data, data[rows].copy(), iter=iter, minit='matrix')
    assert np.array_equal(c2, centroid) and np.array_equal(l2, label), "init-row reconstruction failed"
    _, lab1 = _real_kmeans2(data, data[rows].copy(), iter=1, minit='matrix')
    _, lab2 = _real_kmeans2(data, data[rows].copy(), iter=2, minit='matrix')
    KM_LOG.append(dict(n=data.shape[0], k=int(k), rows=rows.astype(np.int32), labels=label.astype(np.int32),
                       centroid=centroid.astype(np.float32), labels_it1=lab1.astype(np.int32),
                       labels_it2=lab2.astype(np.int32)))
    return centroid, label


aem.kmeans2 = recording_kmeans2
mt.kmeans2 = recording_kmeans2


def f16(x):
    """Round to float16-representable float32 (so the stored float16 copy is exact)."""
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def clip(h, w, c, n_obj, frames, seed):
    cfg = syn.ClipConfig("g", h, w, n_obj, 16, c, frames)
    d = syn.make_clip(cfg, seed)
    return f16(d["emb"]), d["lab"]


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        v = np.asarray(v)
        if v.dtype == np.float32 and k.startswith("in_"):
            assert np.array_equal(v.astype(np.float16).astype(np.float32), v), k
            v = v.astype(np.float16)
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name:34s} {os.path.getsize(path) / 1024:8.1f} KiB")


def km_arrays():
    out = {"km_calls": np.int32(len(KM_LOG))}
    for i, r in enumerate(KM_LOG):
        for k, v in r.items():
            out[f"km{i}_{k}"] = v
    return out


def run_cluster(name, emb, lab, n_obj, ref_ids, q_id, seed, bias=None, ori_size=None, lab_override=None,
                fn=None):
    refs = [torch.from_numpy(emb[i]) for i in ref_ids]
    labs_np = [lab_override[i] if lab_override is not None else syn.one_hot(lab[i], n_obj) for i in ref_ids]
    labs = [torch.from_numpy(l.copy()) for l in labs_np]
    q = torch.from_numpy(emb[q_id])
    b = torch.zeros(n_obj, 1, 1, 1) if bias is None else torch.from_numpy(f16(bias)).view(n_obj, 1, 1, 1)
    KM_LOG.clear()
    np.random.seed(seed)
    fn = fn or aem.global_matching_for_eval_cluster
    out = fn(refs, q, labs, 4, b, ori_size, 1, False, 0)
    save(name, in_ref=np.stack([emb[i] for i in ref_ids]), lab_onehot=np.stack(labs_np).astype(np.float16),
         in_query=emb[q_id], in_bias=f16(b.numpy().reshape(-1)), seed=np.int64(seed),
         ori_size=np.array(ori_size if ori_size else (0, 0), np.int32), out=out.numpy(), **km_arrays())


def main():
    H, W, C = 24, 40, 100

    # ---------------------------------------------------------------- cluster path (a2-a5)
    emb, lab = clip(H, W, C, 3, 5, seed=1)
    run_cluster("cluster_basic_R1_O3", emb, lab, 3, [0], 2, seed=11)
    # cross-check AEM vs the model's own copy (matching.py) -- SURVEY v9
    run_cluster("cluster_basic_R1_O3_mt", emb, lab, 3, [0], 2, seed=11, fn=mt.global_matching_for_eval_cluster)
    emb4, lab4 = clip(H, W, C, 4, 6, seed=2)
    run_cluster("cluster_R3_O4_bias", emb4, lab4, 4, [0, 2, 4], 5, seed=12, bias=[0.25, -0.5, 0.125, 1.0])
    run_cluster("cluster_orisize", emb, lab, 3, [0, 1], 3, seed=13, ori_size=(31, 47))
    # object 0 (background) has no pixel -> sticky K = 0 -> every feature is exactly 1.0
    lab_nobg = lab.copy()
    lab_nobg[lab_nobg == 0] = 1
    run_cluster("cluster_sticky_empty_obj0", emb, lab_nobg, 3, [0], 2, seed=14)
    # an object with fewer pixels than K in front of a large one: K sticks at n_i
    lab_small = np.zeros_like(lab)
    lab_small[:, 3, 5:10] = 1          # object 1: 5 pixels
    lab_small[:, 10:20, 10:30] = 2     # object 2: 200 pixels; background 0 is large and first
    lab_small0 = lab_small.copy()
    lab_small0[lab_small0 == 0] = 9    # background absent from the label set -> rows dropped
    lab_small0[:, 0, 0:3] = 0          # 3 background pixels: K sticks at 3 from the first object on
    run_cluster("cluster_small_first_obj", emb, lab_small0, 3, [0], 2, seed=15)
    run_cluster("cluster_small_mid_obj", emb, lab_small, 3, [0], 2, seed=16)
    # duplicate rows -> empty clusters: centroid_avg has K' < K rows
    emb_dup = emb.copy()
    emb_dup[0, 8:, :, :] = emb_dup[0, 8:9, 0:1, :]
    run_cluster("cluster_duplicates_empty", emb_dup, lab, 3, [0], 2, seed=17)
    # 'uncertain' label 125 matches no object -> all-zero rows are dropped (eval_manager_mm.py:339-349)
    lab_unc = lab.copy()
    lab_unc[:, 5:12, 5:25] = 125
    run_cluster("cluster_uncertain125", emb, lab_unc, 3, [0, 1], 2, seed=18)
    # nothing labelled at all -> early out, [1,h,w,O,1] of ones (AEM:588-589)
    run_cluster("cluster_all_unlabelled", emb, np.full_like(lab, 125), 3, [0], 2, seed=19)

    # ---------------------------------------------------------------- dense path (a6)
    def run_dense(name, emb, lab, n_obj, ref_ids, q_id, bias=None, ori_size=None, fn=None, single=False):
        labs_np = [syn.one_hot(lab[i], n_obj) for i in ref_ids]
        b = torch.zeros(n_obj, 1, 1, 1) if bias is None else torch.from_numpy(f16(bias)).view(n_obj, 1, 1, 1)
        q = torch.from_numpy(emb[q_id])
        if single:
            out = fn(torch.from_numpy(emb[ref_ids[0]]), q, torch.from_numpy(labs_np[0].copy()), 3, b, ori_size, 1, False, 0)
        else:
            out = (fn or aem.global_matching_for_eval)([torch.from_numpy(emb[i]) for i in ref_ids], q,
                                                       [torch.from_numpy(l.copy()) for l in labs_np], 4, b, ori_size, 1, False, 0)
        save(name, in_ref=np.stack([emb[i] for i in ref_ids]), lab_onehot=np.stack(labs_np).astype(np.float16),
             in_query=emb[q_id], in_bias=f16(b.numpy().reshape(-1)),
             ori_size=np.array(ori_size if ori_size else (0, 0), np.int32), out=out.numpy())

    run_dense("dense_R1_O3", emb, lab, 3, [0], 2)
    run_dense("dense_R2_O4_bias_unc", emb4, np.where((lab4 > 0) & (np.indices(lab4.shape)[2] % 7 == 0), 125, lab4),
              4, [1, 3], 5, bias=[0.5, -0.25, 0.0, 2.0])
    run_dense("dense_orisize", emb, lab, 3, [0], 1, ori_size=(29, 51))
    run_dense("dense_all_unlabelled", emb, np.full_like(lab, 125), 3, [0], 1)
    run_dense("dense_train_twin", emb, lab, 3, [0], 2, fn=aem.global_matching, single=True)

    # ---------------------------------------------------------------- k=1 proxy path (a7)
    def run_proxy(name, emb, lab, n_obj, q_id, fn, bias=None):
        proxies = f16(np.random.RandomState(5).rand(n_obj, C).astype(np.float32) * 0.4)
        b = torch.zeros(n_obj, 1, 1, 1) if bias is None else torch.from_numpy(f16(bias)).view(n_obj, 1, 1, 1)
        labs = syn.one_hot(lab[0], n_obj)
        if fn is aem.global_matching_proxy:
            out = fn(torch.from_numpy(proxies), torch.from_numpy(emb[q_id]), torch.from_numpy(labs), 3, b, None, 1, False, 0)
        else:
            out = fn(torch.from_numpy(proxies), torch.from_numpy(emb[q_id]), [torch.from_numpy(labs)], 4, b, None, 1, False, 0)
        save(name, in_proxies=proxies, in_query=emb[q_id], lab_onehot=labs.astype(np.float16)[None],
             in_bias=f16(b.numpy().reshape(-1)), out=out.numpy())

    run_proxy("proxy_eval_O3", emb, lab, 3, 2, mt.global_matching_for_eval_proxy, bias=[0.1, 0.2, -0.3])
    run_proxy("proxy_train_O3", emb, lab, 3, 2, aem.global_matching_proxy)

    # ---------------------------------------------------------------- local path (a8)
    def run_local(name, emb, lab, n_obj, p_id, q_id, mld, bias=None, down=True, fn=None, parallel=True):
        b = torch.zeros(n_obj, 1, 1, 1) if bias is None else torch.from_numpy(f16(bias)).view(n_obj, 1, 1, 1)
        labs = syn.one_hot(lab[p_id], n_obj)
        out = (fn or aem.local_matching)(torch.from_numpy(emb[p_id]), torch.from_numpy(emb[q_id]), torch.from_numpy(labs),
                                         b, list(mld), None, 1, False, down, parallel)
        save(name, in_prev=emb[p_id], in_query=emb[q_id], lab_onehot=labs.astype(np.float16), in_bias=f16(b.numpy().reshape(-1)),
             mld=np.array(mld, np.int32), down=np.int32(down), out=out.numpy())

    MLD = [2, 4, 6, 8, 10, 12]
    run_local("local_down_O3", emb, lab, 3, 1, 2, MLD)
    run_local("local_down_O4_bias", emb4, lab4, 4, 3, 4, MLD, bias=[0.3, -0.2, 0.0, 0.7])
    run_local("local_nodown_O3", emb, lab, 3, 1, 2, [1, 2, 3, 5], down=False)
    # NB allow_parallel=False (AEM:875-919) is not recorded: that loop variant shadows its argument
    # ``x`` with the inner loop index (AEM:911-915) and returns distances to an integer; it is
    # unused (TEST_LOCAL_PARALLEL=True, configs/resnet101_aocnet.py:127).
    # odd map so that int(h/2)+1 rounding is exercised (121x213 -> 61x107 in cfg2)
    embo, labo = clip(25, 37, 36, 3, 3, seed=4)
    run_local("local_down_odd_C36", embo, labo, 3, 0, 1, MLD)
    # local_matching_proxy is fed a per-pixel proxy map (aocnet.py:325)
    prox = f16(np.random.RandomState(6).rand(3, C).astype(np.float32) * 0.4)
    pmap = f16(syn.one_hot(lab[1], 3) @ prox)
    emb_p = emb.copy()
    emb_p[1] = pmap
    run_local("local_proxy_down_O3", emb_p, lab, 3, 1, 2, MLD, fn=aem.local_matching_proxy)

    # ---------------------------------------------------------------- fg2bg (a9)
    d = torch.from_numpy(f16(np.random.RandomState(7).rand(4, 6, 12, 9)))
    save("fg2bg_O4", in_dis=d.numpy(), out=aem.foreground2background(d, 4).numpy())
    save("fg2bg_O1", in_dis=d[:1].numpy(), out=aem.foreground2background(d[:1], 1).numpy())

    # ---------------------------------------------------------------- k=1 proxies / IA head (a10)
    n_obj = 3
    ref_e = [torch.from_numpy(emb[i]).permute(2, 0, 1).unsqueeze(0) for i in (0, 2)]           # [1,C,h,w]
    ref_l = [torch.from_numpy(syn.one_hot(lab[i], n_obj)).permute(2, 0, 1).unsqueeze(1) for i in (0, 2)]  # [O,1,h,w]
    prev_e = torch.from_numpy(emb[3]).permute(2, 0, 1).unsqueeze(0).expand(n_obj, -1, -1, -1)
    prev_l = torch.from_numpy(syn.one_hot(lab[3], n_obj)).permute(2, 0, 1).unsqueeze(1)
    outs = att.calculate_attention_head_for_eval_p_m(ref_e, ref_l, prev_e, prev_l, 1e-5)
    save("attention_head_eval_p_m", in_ref=np.stack([emb[0], emb[2]]), lab_ref=np.stack([lab[0], lab[2]]).astype(np.int16),
         in_prev=emb[3], lab_prev=lab[3].astype(np.int16), n_obj=np.int32(n_obj),
         total=outs[0].numpy(), ref_pos=outs[1].numpy(), ref_neg=outs[2].numpy(), prev_pos=outs[3].numpy(), prev_neg=outs[4].numpy())
    outs_t = att.calculate_attention_head_p_m(ref_e[0].expand(n_obj, -1, -1, -1), ref_l[0], prev_e, prev_l, 1e-5)
    save("attention_head_train_p_m", in_ref=emb[0][None], lab_ref=lab[0][None].astype(np.int16), in_prev=emb[3],
         lab_prev=lab[3].astype(np.int16), n_obj=np.int32(n_obj), total=outs_t[0].numpy())

    # ---------------------------------------------------------------- IA_gate (a11)
    torch.manual_seed(3)
    gate = att.IA_gate(40, 12)
    with torch.no_grad():
        gate.IA.weight.copy_(torch.from_numpy(f16(gate.IA.weight.numpy())))
        gate.IA.bias.copy_(torch.from_numpy(f16(gate.IA.bias.numpy())))
        x = torch.from_numpy(f16(np.random.RandomState(8).randn(3, 12, 9, 11)))
        head = torch.from_numpy(f16(np.random.RandomState(9).randn(3, 40)))
        y = gate(x, head)
    save("ia_gate", in_x=x.numpy(), in_head=head.numpy(), in_w=gate.IA.weight.detach().numpy(),
         in_b=gate.IA.bias.detach().numpy(), out=y.numpy())

    # ---------------------------------------------------------------- conditioning_layer (a12)
    torch.manual_seed(4)
    layer = clb.conditioning_layer(in_dim=24, beta_percentage=0.3)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(torch.from_numpy(f16(p.numpy())))
        z = torch.from_numpy(f16(np.random.RandomState(10).randn(3, 24, 13, 17)))
        clb.mlp_layer = layer.mlp_layer          # the forward references a missing global (CLB:46)
        out = layer(z)
        # intermediates (observed by re-running the reference's own first steps)
        s = layer.phi_layer(z).reshape(3, -1)
        k = int(0.3 * 13 * 17)
        thr = torch.topk(s, k=k, dim=-1, sorted=True)[0][..., -1]
    save("conditioning_layer_4d", in_z=z.numpy(), in_phi_w=layer.phi_layer.weight.detach().numpy().reshape(-1),
         in_phi_b=layer.phi_layer.bias.detach().numpy(), in_mlp_w=layer.mlp_layer.weight.detach().numpy(),
         in_mlp_b=layer.mlp_layer.bias.detach().numpy(), beta=np.float32(0.3), out=out.numpy(),
         scores=s.numpy(), thr=thr.numpy(), mask_count=(s > thr[:, None]).sum(1).numpy().astype(np.int32))


if __name__ == "__main__":
    main()
