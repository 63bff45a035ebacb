#!/usr/bin/env python3
"""More round-2 golden vectors from the reference itself (build container only; see make_golden.py): local matching with
atrous_rate > 1 (AEM:949-959 strided window, AEM:1039 `local_dis // atrous_rate`), with and without the half-resolution path.

    python tests/golden/make_golden_r2b.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import aem, syn, f16, save, clip  # noqa: E402


def run_local(name, emb, lab, n_obj, p_id, q_id, mld, rate, bias=None, down=True, f16mode=False):
    b = torch.zeros(n_obj, 1, 1, 1) if bias is None else torch.from_numpy(f16(bias)).view(n_obj, 1, 1, 1)
    labs = syn.one_hot(lab[p_id], n_obj)
    out = aem.local_matching(torch.from_numpy(emb[p_id]), torch.from_numpy(emb[q_id]), torch.from_numpy(labs), b, list(mld), None, rate, f16mode, down, True)
    save(name, in_prev=emb[p_id], in_query=emb[q_id], lab_onehot=labs.astype(np.float16), in_bias=f16(b.numpy().reshape(-1)),
         mld=np.array(mld, np.int32), down=np.int32(down), atrous_rate=np.int32(rate), float16=np.int32(f16mode), out=out.numpy())


def main():
    emb, lab = clip(24, 40, 100, 3, 5, seed=1)
    run_local("local_atrous2_down_O3", emb, lab, 3, 1, 2, [2, 4, 6, 8, 10, 12], 2, bias=[0.1, -0.2, 0.3])
    run_local("local_atrous3_nodown_O3", emb, lab, 3, 1, 2, [2, 5, 7, 11], 3, down=False)      # pad = 11 - 11 % 3 = 9; rings 0, 1, 2 | 3
    run_local("local_atrous2_fp16_down_O3", emb, lab, 3, 1, 2, [2, 4, 6, 8, 10, 12], 2, f16mode=True)


if __name__ == "__main__":
    main()
