"""Adaptive LayerNorm layers (reference diffnext/models/normalization.py:24-46).

state_dict keys: `proj.{weight,bias}` (+ `lora.weight` when rank is set). On the MI355X inference
path these layers are never called one by one: the engine concatenates every block's `proj` into
a single [(3*depth+2)D, D] GEMM per diffusion step and fuses the modulation into `nova_row_norm`.
"""
from typing import Tuple

import torch
from torch import nn
from torch.nn import functional as F


class AdaLayerNormZero(nn.Module):
    """LN (no affine) modulated by statistics regressed from a condition; extra stats are returned."""

    def __init__(self, dim, rank=None, num_stats=2, eps=1e-6):
        super().__init__()
        self.lora = nn.Linear(dim, rank, bias=False) if rank else nn.Identity()
        self.proj = nn.Linear(rank if rank else dim, num_stats * dim)
        self.norm = nn.LayerNorm(dim, eps, elementwise_affine=False) if eps else nn.Identity()
        self.activation, self.num_stats = nn.SiLU(), num_stats

    def forward(self, x, z) -> Tuple[torch.Tensor, Tuple[torch.Tensor]]:
        stats = self.proj(self.lora(F.silu(z))).chunk(self.num_stats, dim=-1)
        scale, shift = stats[0], stats[1]
        return self.norm(x) * (1 + scale) + shift, stats[2:]


class AdaLayerNorm(AdaLayerNormZero):
    """Two-statistic variant returning only the modulated tensor (video mixer, T > 1)."""

    def __init__(self, dim, rank=None, eps=1e-6):
        super().__init__(dim, rank, num_stats=2, eps=eps)

    def forward(self, x, z) -> torch.Tensor:
        return super().forward(x, z)[0]
