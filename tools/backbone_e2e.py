#!/usr/bin/env python3
"""Image-level end-to-end figure of SURVEY.md 8(d) ("reported separately"): a random-weight ResNet101-DeepLabv3+ feature extractor + the
semantic-embedding head of AOCNet (aocnet.py:19-25) in PLAIN PyTorch-ROCm (MIOpen convolutions, fp32, frozen BatchNorm) in front of this
repo's hot path, on N(0,1) 481x849 frames.  MEASUREMENT ONLY: the backbone is out of scope (north_star: "host code stays Python on
PyTorch-ROCm for the ResNet101-DeepLabv3+ backbone"); nothing of this file is product code, and no backbone kernel of this repo exists.

The architecture is the standard one the reference configures (networks/deeplab/deeplab.py:10-33: ResNet101 at output stride 16 with the
last stage dilated, ASPP with rates 6/12/18 + image pooling -> 256, decoder: 48-channel low-level projection, x4 bilinear up-sample, two
3x3 convolutions -> 256 channels at stride 4), written from the published DeepLabv3+ description (torchvision is not installed here).

Prints one JSON object: backbone ms per frame, hot-path ms per frame at the same pool size, and the sequential end-to-end frames/s.
    python tools/backbone_e2e.py [--frames 20] [--pool-frames 6]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class FrozenBN(nn.Module):
    """BatchNorm with fixed statistics folded into scale / shift (MODEL_FREEZE_BN = True, configs/resnet101_aocnet.py:79)."""

    def __init__(self, c):
        super().__init__()
        self.register_buffer("scale", torch.ones(1, c, 1, 1))
        self.register_buffer("shift", torch.zeros(1, c, 1, 1))

    def forward(self, x):
        return x * self.scale + self.shift


class Bottleneck(nn.Module):
    def __init__(self, cin, planes, stride=1, dilation=1, down=False):
        super().__init__()
        self.c1, self.b1 = nn.Conv2d(cin, planes, 1, bias=False), FrozenBN(planes)
        self.c2, self.b2 = nn.Conv2d(planes, planes, 3, stride, dilation, dilation, bias=False), FrozenBN(planes)
        self.c3, self.b3 = nn.Conv2d(planes, planes * 4, 1, bias=False), FrozenBN(planes * 4)
        self.down = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride, bias=False), FrozenBN(planes * 4)) if down else None

    def forward(self, x):
        r = x if self.down is None else self.down(x)
        y = F.relu(self.b1(self.c1(x)))
        y = F.relu(self.b2(self.c2(y)))
        return F.relu(self.b3(self.c3(y)) + r)


def stage(cin, planes, n, stride, dilation):
    blocks = [Bottleneck(cin, planes, stride, dilation, down=True)]
    blocks += [Bottleneck(planes * 4, planes, 1, dilation) for _ in range(n - 1)]
    return nn.Sequential(*blocks)


class DeepLabV3Plus(nn.Module):
    def __init__(self, emb_dim=100):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), FrozenBN(64), nn.ReLU(True), nn.MaxPool2d(3, 2, 1))
        self.l1, self.l2 = stage(64, 64, 3, 1, 1), stage(256, 128, 4, 2, 1)
        self.l3, self.l4 = stage(512, 256, 23, 2, 1), stage(1024, 512, 3, 1, 2)                 # output stride 16
        self.aspp = nn.ModuleList([nn.Sequential(nn.Conv2d(2048, 256, 1 if r == 1 else 3, padding=0 if r == 1 else r, dilation=r, bias=False), FrozenBN(256),
                                                 nn.ReLU(True)) for r in (1, 6, 12, 18)])
        self.pool = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(2048, 256, 1, bias=False), FrozenBN(256), nn.ReLU(True))
        self.proj = nn.Sequential(nn.Conv2d(1280, 256, 1, bias=False), FrozenBN(256), nn.ReLU(True))
        self.low = nn.Sequential(nn.Conv2d(256, 48, 1, bias=False), FrozenBN(48), nn.ReLU(True))
        self.last = nn.Sequential(nn.Conv2d(304, 256, 3, padding=1, bias=False), FrozenBN(256), nn.ReLU(True),
                                  nn.Conv2d(256, 256, 3, padding=1, bias=False), FrozenBN(256), nn.ReLU(True))
        # AOCNet.semantic_embedding (aocnet.py:19-25): depthwise 3x3, GroupNorm(32), ReLU, 1x1 -> emb_dim, GroupNorm(25), ReLU
        self.emb = nn.Sequential(nn.Conv2d(256, 256, 3, padding=1, groups=256), nn.GroupNorm(32, 256), nn.ReLU(True),
                                 nn.Conv2d(256, emb_dim, 1), nn.GroupNorm(25, emb_dim), nn.ReLU(True))

    def forward(self, img):
        x = self.stem(img)
        low = self.l1(x)
        x = self.l4(self.l3(self.l2(low)))
        a = [m(x) for m in self.aspp] + [F.interpolate(self.pool(x), size=x.shape[2:], mode="bilinear", align_corners=True)]
        x = self.proj(torch.cat(a, 1))
        x = F.interpolate(x, size=low.shape[2:], mode="bilinear", align_corners=True)
        x = self.last(torch.cat([x, self.low(low)], 1))
        return self.emb(x), low                                # [1, emb_dim, h, w] at stride 4, low-level features


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--pool-frames", type=int, default=6, help="reference pool size of the hot-path frames (the cfg2 clip's mean is 6.4)")
    args = ap.parse_args()
    import aoc_amd
    from aoc_amd import hotpath, synthetic as syn
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = DeepLabV3Plus().to(dev).eval()
    img = torch.randn(1, 3, 481, 849, device=dev)
    cfg = syn.CONFIGS["cfg2"]
    mc = hotpath.MatchingConfig()
    O, R = cfg.n_obj, args.pool_frames
    with torch.no_grad():
        e, _ = net(img)
        assert tuple(e.shape) == (1, 100, cfg.h, cfg.w), e.shape
        for _ in range(3):
            net(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.frames):
            net(img)
        torch.cuda.synchronize()
        backbone_ms = (time.perf_counter() - t0) / args.frames * 1e3

        # hot path at R pool frames on synthetic feature clips (the backbone's random-weight outputs saturate the matching: SURVEY v15)
        clip = syn.make_clip(cfg, 3, frames=R * mc.MEM_EVERY + 2)
        emb = torch.from_numpy(clip["emb"]).to(dev)
        lab = torch.from_numpy(np.stack([syn.one_hot(l, O) for l in clip["lab"]])).to(dev)
        ref_emb, ref_lab = emb[0:R * mc.MEM_EVERY:mc.MEM_EVERY].contiguous(), lab[0:R * mc.MEM_EVERY:mc.MEM_EVERY].contiguous()
        counts = [int(ref_lab[..., o].sum().item()) for o in range(O)]
        rows = np.zeros((O, 16), np.int32)
        for o, r in enumerate(syn.kmeans_init_rows(5, counts, 16)):
            rows[o, :len(r)] = r
        init = torch.from_numpy(rows).to(dev)
        runner = hotpath.FrameRunner(mc, cfg.h, cfg.w, cfg.c, O, R, dev)
        bias = torch.zeros(O, device=dev)
        tq = R * mc.MEM_EVERY + 1

        def hot():
            ahead = hotpath.launch_cluster_proxies(mc, ref_emb, ref_lab, init)
            return runner(ref_emb, ref_lab, emb[tq - 1], lab[tq - 1], emb[tq], bias, ahead, pool_key=ref_emb.shape[0])

        for _ in range(3):
            hot()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.frames):
            hot()
        torch.cuda.synchronize()
        hot_ms = (time.perf_counter() - t0) / args.frames * 1e3
        for _ in range(2):
            net(img); hot()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.frames):
            net(img)
            hot()
        torch.cuda.synchronize()
        e2e_ms = (time.perf_counter() - t0) / args.frames * 1e3
    print(json.dumps(dict(
        what="image-level end to end, ONE sequence, frames one after another: plain-PyTorch random-weight ResNet101-DeepLabv3+ + embedding head "
             "(MIOpen fp32) -> this repo's matching hot path (k-means chain on the frame's own stream, no cross-sequence overlap, no calibration gates)",
        input="N(0,1) 481x849 frame -> 121x213 stride-4 map", pool_frames=R, frames=args.frames,
        backbone_ms_per_frame=round(backbone_ms, 3), hot_path_ms_per_frame=round(hot_ms, 3), end_to_end_ms_per_frame=round(e2e_ms, 3),
        end_to_end_frames_per_s=round(1e3 / e2e_ms, 2), hot_path_share=round(hot_ms / (backbone_ms + hot_ms), 3))))


if __name__ == "__main__":
    main()
