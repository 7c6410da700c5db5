"""Adaptive LayerNorm layers: the module surface of reference diffnext/models/normalization.py:24-46.

state_dict keys: `proj.{weight,bias}` (+ `lora.weight` when a rank is given). On the MI355X inference path these
layers are never called one by one: the engine concatenates every block's `proj` into a single
[(3 * depth + 2) D, D] GEMM per diffusion step and fuses the modulation into `nova_row_norm`.
"""
from typing import Tuple

import torch
from torch import nn
from torch.nn import functional as F

from .. import _backend
from .. import _torch_ops as ops


class AdaLayerNormZero(nn.Module):
    """Affine-free LayerNorm whose scale and shift are regressed from a condition z.

    `proj(lora(SiLU(z)))` yields `num_stats` chunks of width `dim`: the first two modulate, the rest (gates) are
    handed back to the caller.
    """

    def __init__(self, dim, rank=None, num_stats=2, eps=1e-6):
        super().__init__()
        self.num_stats = num_stats
        self.activation = nn.SiLU()
        if rank:
            self.lora = nn.Linear(dim, rank, bias=False)
            self.proj = nn.Linear(rank, num_stats * dim)
        else:
            self.lora = nn.Identity()
            self.proj = nn.Linear(dim, num_stats * dim)
        self.norm = nn.LayerNorm(dim, eps, elementwise_affine=False) if eps else nn.Identity()

    def statistics(self, z):
        return self.proj(self.lora(_backend.train_activation(z, 2, F.silu))).chunk(self.num_stats, dim=-1)

    def forward(self, x, z) -> Tuple[torch.Tensor, Tuple[torch.Tensor]]:
        scale, shift, *rest = self.statistics(z)
        if isinstance(self.norm, nn.LayerNorm) and _backend.train_norm_supported(x, scale=scale, shift=shift):
            # training on the GPU: affine-free LayerNorm + modulate as one HIP row kernel each way (csrc/rownorm_bwd.hip)
            return _backend.autograd().fused_norm(x, scale=scale, shift=shift, eps=self.norm.eps), tuple(rest)
        return ops.adaln_modulate(self.norm(x), scale, shift), tuple(rest)


class AdaLayerNorm(AdaLayerNormZero):
    """Two-statistic variant returning only the modulated tensor (video mixer, T > 1)."""

    def __init__(self, dim, rank=None, eps=1e-6):
        super().__init__(dim, rank, num_stats=2, eps=eps)

    def forward(self, x, z) -> torch.Tensor:
        modulated, _ = AdaLayerNormZero.forward(self, x, z)
        return modulated
