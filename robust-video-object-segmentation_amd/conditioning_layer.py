"""Drop-in mirror of AOC-Net/conditioning_layer.py ("CL", == networks/aoc/conditioning_layer.py
"CLB":6-48) and of ``conditioning_block`` (CLB:50-86) on the HIP library.

Parameter names (``phi_layer``, ``mlp_layer``, ``CL_1``, ``CL_2``, ``CL_3``) are the reference's, so a
reference ``state_dict`` loads.  The reference code is not executable as shipped;
the repairs are the minimal ones, spelled out in DESIGN.md:
  * ``self.`` in front of ``mlp_layer`` / ``CL_1..3``;
  * ``conditioning_block`` feeds ``CL_2`` / ``CL_3`` with 2-D tensors that ``Conv2d`` rejects (and, as
    [N,D,1,1], ``k = int(beta*1*1) = 0`` raises): a vector input is treated as the H = W = 1 limit of
    paper Eq. 7 with the gate identically 1, i.e. ``CL(v) = mlp_layer(v)``;
  * ``CL_3`` is sized by ``proxy_dim`` (its input is the [O, 400] IA head).
"""
import torch
from torch import nn

from . import ops


class conditioning_layer(nn.Module):
    """Paper Eq. 7, CL:6-45."""

    def __init__(self, in_dim=256, beta_percentage=0.3):
        super(conditioning_layer, self).__init__()
        self.beta_percentage = beta_percentage
        kernel_size = 1
        self.phi_layer = nn.Conv2d(in_dim, 1, kernel_size=kernel_size, stride=1, padding=int((kernel_size - 1) / 2))
        self.mlp_layer = nn.Linear(in_dim, in_dim)
        nn.init.kaiming_normal_(self.phi_layer.weight, mode='fan_out', nonlinearity='relu')

    def forward(self, z_in):
        ops.inference_only("conditioning_layer", z_in, *self.parameters())
        if z_in.dim() == 2:                                   # vector input (conditioning_block repair)
            return ops.linear(z_in, self.mlp_layer.weight.detach(), self.mlp_layer.bias.detach())
        beta_rank = int(self.beta_percentage * z_in.size()[-1] * z_in.size()[-2])      # CL:32
        if beta_rank < 1:
            raise IndexError("conditioning_layer: beta_rank == 0 (the reference fails at beta_val[..., -1])")
        gap = ops.cond_gate_pool(z_in, self.phi_layer.weight.detach().reshape(-1), self.phi_layer.bias.detach(), beta_rank)
        return ops.linear(gap, self.mlp_layer.weight.detach(), self.mlp_layer.bias.detach())   # CL:46


class conditioning_block(nn.Module):
    """Paper Eq. 5, CLB:50-86."""

    def __init__(self, in_dim=256, proxy_dim=400, beta_percentage=0.3, attention_dim=None):
        super(conditioning_block, self).__init__()
        # decoding_module.py:27-50 constructs the blocks with ``attention_dim=`` (a keyword the reference's own class rejects, CLB:54):
        # accepted here as the width of the IA head, i.e. as ``proxy_dim``
        if attention_dim is not None:
            proxy_dim = attention_dim
        self.CL_1 = conditioning_layer(in_dim, beta_percentage)
        self.CL_2 = conditioning_layer(in_dim, beta_percentage)
        self.CL_3 = conditioning_layer(proxy_dim, 1)
        self.mlp_layer = nn.Linear(in_dim * 2 + proxy_dim, in_dim)

    def forward(self, x, proxy_IA_head):
        ops.inference_only("conditioning_block", x, proxy_IA_head, *self.parameters())
        beta_rank = int(self.CL_1.beta_percentage * x.size()[-1] * x.size()[-2])       # CL:32
        if beta_rank < 1:
            raise IndexError("conditioning_layer: beta_rank == 0 (the reference fails at beta_val[..., -1])")
        # CLB:68 (plane means of x) and CLB:72 / CL:28-45 (gate + masked pooling): the plane sums come out of the pass that computes the scores
        gap, px1 = ops.cond_gate_pool(x, self.CL_1.phi_layer.weight.detach().reshape(-1), self.CL_1.phi_layer.bias.detach(), beta_rank,
                                      want_plane_mean=True)
        # CLB:69 (inter-object delta), the three mlp_layer products (CLB:72-78, CL:46) and the concatenation (CLB:80) in one launch
        code = ops.cond_codes(gap, px1, proxy_IA_head,
                              self.CL_1.mlp_layer.weight.detach(), self.CL_1.mlp_layer.bias.detach(),
                              self.CL_2.mlp_layer.weight.detach(), self.CL_2.mlp_layer.bias.detach(),
                              self.CL_3.mlp_layer.weight.detach(), self.CL_3.mlp_layer.bias.detach())
        return ops.film_scale(x, code, self.mlp_layer.weight.detach(), self.mlp_layer.bias.detach())   # CLB:81-84
