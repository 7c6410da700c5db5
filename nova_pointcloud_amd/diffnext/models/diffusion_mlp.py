"""Per-token diffusion MLP (reference diffnext/models/diffusion_mlp.py).

Projector :26-36, DiffusionBlock :39-53 (AdaLN-Zero residual MLP), TimeCondEmbed :56-75,
DiffusionMLP :78-99 — same constructors and state_dict keys. The PyTorch definitions below serve
CPU tensors and training; in generation on an MI355X the engine runs all diffusion steps of an
AR step through `nova_decoder_denoise` (one concatenated AdaLN GEMM per step, fused
LN-modulate / gate-LN-residual row kernels, head + CFG + Euler fused) and a direct module call
with device tensors goes through `engine.decoder_forward`.
"""
import torch
from torch import nn
from torch.nn import functional as F
from torch.utils.checkpoint import checkpoint

from .. import _backend
from .embeddings import PatchEmbed
from .normalization import AdaLayerNormZero
from .vision_transformer import use_hip


class Projector(nn.Module):
    """Linear -> SiLU -> Linear."""

    def __init__(self, dim, mlp_dim=None, out_dim=None):
        super().__init__()
        self.fc1 = nn.Linear(dim, mlp_dim or dim)
        self.fc2 = nn.Linear(mlp_dim or dim, out_dim or dim)
        self.activation = nn.SiLU()

    def forward(self, x) -> torch.Tensor:
        return self.fc2(F.silu(self.fc1(x)))


class DiffusionBlock(nn.Module):
    """x + gate * LN(proj(AdaLN(x, z)))."""

    def __init__(self, dim):
        super().__init__()
        self.dim, self.mlp_checkpointing = dim, False
        self.norm1 = AdaLayerNormZero(dim, num_stats=3, eps=1e-6)
        self.proj, self.norm2 = Projector(dim, dim, dim), nn.LayerNorm(dim)

    def forward(self, x, z) -> torch.Tensor:
        if self.mlp_checkpointing and x.requires_grad:
            h, (gate,) = checkpoint(self.norm1, x, z, use_reentrant=False)
            return self.norm2(checkpoint(self.proj, h, use_reentrant=False)) * gate + x
        h, (gate,) = self.norm1(x, z)
        return self.norm2(self.proj(h)) * gate + x


class TimeCondEmbed(nn.Module):
    """Sinusoidal timestep features and the encoder condition, each through a Projector, summed."""

    def __init__(self, cond_dim, embed_dim, freq_dim=256):
        super().__init__()
        self.timestep_proj = Projector(freq_dim, embed_dim, embed_dim)
        self.condition_proj = Projector(cond_dim, embed_dim, embed_dim)
        self.freq_dim, self.time_freq = freq_dim, None

    def get_freq_embed(self, timestep, dtype) -> torch.Tensor:
        if self.time_freq is None or self.time_freq.device != timestep.device:
            half = self.freq_dim // 2
            k = torch.arange(half, dtype=torch.float32, device=timestep.device)
            self.time_freq = k.mul(-9.210340371976184 / half).exp().unsqueeze(0)  # exp(-ln(1e4) k / half)
        ang = timestep.unsqueeze(-1).float() * self.time_freq
        return torch.cat([ang.cos(), ang.sin()], dim=-1).to(dtype=dtype)

    def forward(self, timestep, z) -> torch.Tensor:
        t = self.timestep_proj(self.get_freq_embed(timestep, z.dtype))
        return self.condition_proj(z) + (t.unsqueeze(1) if t.dim() == 2 else t)


class DiffusionMLP(nn.Module):
    def __init__(self, depth, embed_dim, cond_dim, patch_size=2, image_dim=4):
        super().__init__()
        self.patch_embed = PatchEmbed(image_dim, embed_dim, patch_size)
        self.time_cond_embed = TimeCondEmbed(cond_dim, embed_dim)
        self.blocks = nn.ModuleList(DiffusionBlock(embed_dim) for _ in range(depth))
        self.norm = AdaLayerNormZero(embed_dim, num_stats=2, eps=1e-6)
        self.head = nn.Linear(embed_dim, patch_size**2 * image_dim)

    def forward(self, x, timestep, z, pred_ids=None) -> torch.Tensor:
        """x [S,C,H,W] noisy canvas, z [S,N,D] condition; only rows `pred_ids` [S,n,1] are computed,
        the other rows of the result echo patchify(x)."""
        if use_hip(z):
            return _backend.engine().decoder_forward(self, x, timestep, z, pred_ids)
        tokens = self.patch_embed(x)
        echo = None if pred_ids is None else self.patch_embed.patchify(x)
        if pred_ids is not None:
            tokens = tokens.gather(1, pred_ids.expand(-1, -1, tokens.size(-1)))
            z = z.gather(1, pred_ids.expand(-1, -1, z.size(-1)))
        z = self.time_cond_embed(timestep, z)
        for blk in self.blocks:
            tokens = blk(tokens, z)
        out = self.head(self.norm(tokens, z)[0])
        return out if pred_ids is None else echo.scatter(1, pred_ids.expand(-1, -1, out.size(-1)), out)
