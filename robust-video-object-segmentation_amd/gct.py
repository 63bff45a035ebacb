"""Decoder-side streams next to the FiLM gates: drop-in mirrors of ``networks/layers/gct.py:GCT`` (17-36)
and of ``CalibrationDecoding.IA_logit`` (networks/aoc/decoding_module.py:151-160) on the HIP library.  Parameter names
(``alpha``, ``gamma``, ``beta``) are the reference's, so a reference ``state_dict`` loads."""
import torch
from torch import nn

from . import ops


class GCT(nn.Module):
    """Gated channel transformation, gct.py:7-36: y = x * (1 + tanh(embedding * norm + beta))."""

    def __init__(self, num_channels, epsilon=1e-5, mode='l2', after_relu=False):
        super(GCT, self).__init__()
        self.alpha = nn.Parameter(torch.ones(1, num_channels, 1, 1))
        self.gamma = nn.Parameter(torch.zeros(1, num_channels, 1, 1))
        self.beta = nn.Parameter(torch.zeros(1, num_channels, 1, 1))
        self.epsilon = epsilon
        self.mode = mode
        self.after_relu = after_relu

    def forward(self, x):
        ops.inference_only("GCT", x, self.alpha, self.gamma, self.beta)
        if self.mode == 'l2':
            sums = ops.plane_reduce(x, 1)                                   # gct.py:19
            l1 = False
        elif self.mode == 'l1':
            sums = ops.plane_reduce(x, 0 if self.after_relu else 2)         # gct.py:23-27
            l1 = True
        else:
            raise ValueError("Unknown mode!")                               # gct.py:29-31 prints and exits
        gate = ops.gct_gate(sums, self.alpha.detach(), self.gamma.detach(), self.beta.detach(), self.epsilon, l1)
        return ops.channel_scale(x, gate)                                   # gct.py:33-35


class Bottleneck(nn.Module):
    """networks/layers/gct.py:38-90 with the reference's parameter names (``GCT1``, ``conv1..3``, ``bn1..3``, ``downsample``).  The
    convolutions are ordinary PyTorch-ROCm modules (MIOpen; out of scope), everything between them runs in the HIP library: the GCT gate
    and GroupNorm + ReLU (+ the residual add in front of the last ReLU) fused into two streams per normalisation."""

    def __init__(self, inplanes, outplanes, stride=1, dilation=1):
        super(Bottleneck, self).__init__()
        expansion = 4
        planes = int(outplanes / expansion)
        self.GCT1 = GCT(inplanes)
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.GroupNorm(32, planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, dilation=dilation, padding=dilation, bias=False)
        self.bn2 = nn.GroupNorm(32, planes)
        self.conv3 = nn.Conv2d(planes, planes * expansion, kernel_size=1, bias=False)
        self.bn3 = nn.GroupNorm(32, planes * expansion)
        if stride != 1 or inplanes != planes * expansion:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * expansion, kernel_size=1, stride=stride, bias=False),
                                            nn.GroupNorm(32, planes * expansion))
        else:
            self.downsample = None
        self.stride = stride
        self.dilation = dilation
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    @staticmethod
    def _gn(bn, x, residual=None, relu=True):
        return ops.groupnorm_relu(x, bn.num_groups, bn.weight.detach(), bn.bias.detach(), bn.eps, residual, relu)

    def forward(self, x):
        ops.inference_only("Bottleneck", x, *self.parameters())
        out = self._gn(self.bn1, self.conv1(self.GCT1(x)))                  # gct.py:69-72
        out = self._gn(self.bn2, self.conv2(out))                           # :74-76
        residual = x
        if self.downsample is not None:
            residual = self._gn(self.downsample[1], self.downsample[0](x), relu=False)     # :81-82
        return self._gn(self.bn3, self.conv3(out), residual=residual)       # :78-79, 84-85


def IA_logit(x, IA_head, IA_final):
    """decoding_module.py:151-160: per-object 1x1 convolution whose C weights and bias come from ``IA_final(IA_head)``
    (an ``nn.Linear(head_dim, C + 1)``).  x [N, C, H, W] -> logit [N, 1, H, W]."""
    ops.inference_only("IA_logit", x, IA_head, IA_final.weight, IA_final.bias)
    out = ops.linear(IA_head, IA_final.weight.detach(), IA_final.bias.detach())      # :154 [N, C + 1]
    return ops.object_logit(x, out)
