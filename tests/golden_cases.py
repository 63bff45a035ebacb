"""Helpers shared by the CPU (oracle) and GPU (product) tests of the round-4 goldens (tests/golden/make_golden_r4.py)."""
import numpy as np
import torch

T = torch.from_numpy
FRAME_CASES = ["frame_R1_O2", "frame_R3_O4_bias", "frame_R2_O3_absent_unc", "frame_R2_O3_nobg"]


def frame_inputs(g):
    prehead = dict(conv_w=T(g["prehead_conv_w"]), conv_b=T(g["prehead_conv_b"]), gn_w=T(g["prehead_gn_w"]), gn_b=T(g["prehead_gn_b"]),
                   groups=int(g["prehead_groups"]), eps=float(g["prehead_eps"]))
    return dict(ref_emb=T(g["in_ref"]), ref_labels_full=T(g["ref_labels_full"].astype(np.int64)), prev_emb=T(g["in_prev"]),
                prev_label_full=T(g["prev_label_full"].astype(np.int64)), cur_emb=T(g["in_cur"]), n_obj=int(g["n_obj"]),
                bg_bias=float(g["bg_bias"]), fg_bias=float(g["fg_bias"]), prehead=prehead, matching_background=bool(g["background"]))


def replay_eval_loop(g, policy, to_dev=lambda t: t):
    """Drives a MemoryPolicy (oracle or product) with the golden's scripted soft-max maps and ground truth exactly as
    eval_manager_mm.py:196-361 drives its lists; yields per frame what the model would be handed."""
    n = int(g["n_frames"])
    gt = {int(t): g[f"gt{int(t)}"].astype(np.int64) for t in g["gt_frames"]}
    emb = lambda t: to_dev(torch.full((1, 1, 4), float(t)))
    for t in range(n):
        ref_frames = [int(e.reshape(-1)[0].item()) for e in policy.ref_embeddings]
        ref_masks = [m.reshape(m.shape[-2], m.shape[-1]).cpu().to(torch.int64).numpy() for m in policy.ref_mask_confident]
        prev_frame = -1 if policy.prev_embedding is None else int(policy.prev_embedding.reshape(-1)[0].item())
        prev_mask = None if policy.prev_mask is None else policy.prev_mask.cpu().to(torch.int64).numpy()
        saved = None
        if t == 0:
            policy.start(emb(0), to_dev(T(gt[0])))
        else:
            label, _, _ = policy.update(emb(t), to_dev(T(g["probs"][t - 1].copy())), to_dev(T(gt[t])) if t in gt else None)
            saved = label.cpu().to(torch.int64).numpy()
        yield t, ref_frames, ref_masks, prev_frame, prev_mask, saved


def check_eval_loop(g, policy, to_dev=lambda t: t):
    H, W = g["probs"].shape[-2:]
    for t, ref_frames, ref_masks, prev_frame, prev_mask, saved in replay_eval_loop(g, policy, to_dev):
        assert ref_frames == g[f"f{t}_ref_frames"].tolist(), f"frame {t}: pool membership"
        want = g[f"f{t}_ref_masks"]
        assert len(ref_masks) == want.shape[0]
        for r, m in enumerate(ref_masks):
            assert np.array_equal(m, want[r]), f"frame {t}: confident reference mask {r}"
        assert prev_frame == int(g[f"f{t}_prev_frame"])
        if t > 0:
            assert np.array_equal(prev_mask.reshape(H, W), g[f"f{t}_prev_mask"]), f"frame {t}: previous mask"
            assert np.array_equal(saved.reshape(H, W), g["saved_labels"][t - 1]), f"frame {t}: saved label map"


BLOCK_CASES = ["conditioning_block_injected_small", "conditioning_block_injected_wide", "conditioning_block_injected_one_object"]


def block_weights(g):
    """The oracle's weight dict from a round-6 conditioning_block golden (tests/golden/make_golden_r6.py stores the reference module's state_dict)."""
    sd = {k[2:].replace("__", "."): T(g[k]) for k in list(g.keys()) if k.startswith("w_")}
    return {"CL_1.phi_w": sd["CL_1.phi_layer.weight"].reshape(-1), "CL_1.phi_b": sd["CL_1.phi_layer.bias"],
            "CL_1.mlp_w": sd["CL_1.mlp_layer.weight"], "CL_1.mlp_b": sd["CL_1.mlp_layer.bias"],
            "CL_2.mlp_w": sd["CL_2.mlp_layer.weight"], "CL_2.mlp_b": sd["CL_2.mlp_layer.bias"],
            "CL_3.mlp_w": sd["CL_3.mlp_layer.weight"], "CL_3.mlp_b": sd["CL_3.mlp_layer.bias"],
            "mlp_w": sd["mlp_layer.weight"], "mlp_b": sd["mlp_layer.bias"]}, sd
