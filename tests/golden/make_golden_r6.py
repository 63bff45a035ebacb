#!/usr/bin/env python3
"""Round 6: golden vectors of ``conditioning_block.forward`` (row a13), recorded by RUNNING THE REFERENCE'S OWN FORWARD.

CLB = /root/reference/AOC-Net/complete_project/AOCNet/networks/aoc/conditioning_layer.py.  CLB:66-86 cannot run as shipped: it calls
``CL_1`` / ``CL_2`` / ``CL_3`` (and ``conditioning_layer.forward`` calls ``mlp_layer``, CLB:46) as module GLOBALS, and it hands ``CL_2`` /
``CL_3`` 2-D tensors that their Conv2d / top-k cannot take.  Python resolves a global at call time in the defining module's namespace, so the
reference's forward executes UNMODIFIED once those four names exist there:

  * ``CL_1``      = the block's own reference ``conditioning_layer`` (4-D input: the reference's code, CLB:24-48, every line of it);
  * ``mlp_layer`` = ``CL_1.mlp_layer`` (the layer's own MLP: the only reading under which CLB:46 refers to a layer that exists);
  * ``CL_2`` / ``CL_3`` = the documented vector repair ``v -> CL_k.mlp_layer(v)`` (oracle/calibration.py header): the only
    part of the block that stays pinned by the restatement alone.

Recorded with the reference's forward: CLB:68-69 (plane means, inter-object delta), CLB:72 through the real layer, CLB:81-84 (concatenation order,
``1 + tanh``, the broadcast product).  Runs only in the build container; only the ``.npz`` files travel.

    python tests/golden/make_golden_r6.py
"""
import importlib.util
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/AOC-Net"
warnings.simplefilter("ignore")
torch.set_num_threads(4)


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def f16(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def main():
    clb = load("ref_clb_r6", f"{REF}/complete_project/AOCNet/networks/aoc/conditioning_layer.py")
    cases = [("conditioning_block_injected_small", 3, 24, 40, 13, 17, 0.3, 21),
             ("conditioning_block_injected_wide", 4, 32, 128, 19, 27, 0.3, 22),      # proxy head wider than the activation, as in decoding_module.py:55-58
             ("conditioning_block_injected_one_object", 1, 16, 12, 9, 11, 0.5, 23)]  # one sample: x_delta is all zeros (sum - itself)
    for name, n, c, p, h, w, beta, seed in cases:
        torch.manual_seed(seed)
        blk = clb.conditioning_block(in_dim=c, proxy_dim=p, beta_percentage=beta)
        with torch.no_grad():
            for prm in blk.parameters():
                prm.copy_(torch.from_numpy(f16(prm.numpy())))
            rs = np.random.RandomState(seed)
            x = torch.from_numpy(f16(rs.randn(n, c, h, w)))
            head = torch.from_numpy(f16(rs.randn(n, p)))
            clb.CL_1 = blk.CL_1
            clb.mlp_layer = blk.CL_1.mlp_layer
            clb.CL_2 = lambda v, _b=blk: _b.CL_2.mlp_layer(v)
            clb.CL_3 = lambda v, _b=blk: _b.CL_3.mlp_layer(v)
            out = blk(x, head)                                  # the reference's forward, CLB:66-86
            cl1 = blk.CL_1(x)                                   # intermediate: the intra-object code (CLB:72), the reference layer itself
        sd = {k: v.detach().numpy() for k, v in blk.state_dict().items()}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), in_x=x.numpy().astype(np.float16), in_head=head.numpy().astype(np.float16), beta=np.float32(beta),
                            **{"w_" + k.replace(".", "__"): v for k, v in sd.items()}, out=out.numpy(), cl1=cl1.numpy())
        print(name, tuple(out.shape), float(out.abs().max()))


if __name__ == "__main__":
    main()
