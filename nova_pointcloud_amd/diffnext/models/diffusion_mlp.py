"""Per-token diffusion MLP: the module surface of reference diffnext/models/diffusion_mlp.py.

Projector :26-36, DiffusionBlock :39-53, TimeCondEmbed :56-75, DiffusionMLP :78-99 - same constructors and
state_dict keys. These PyTorch definitions serve CPU tensors and training. In generation on an MI355X the engine
runs all diffusion steps of an AR step through `nova_decoder_denoise` (one concatenated AdaLN GEMM per step, fused
LN-modulate / gate-LN-residual row kernels, head + CFG + Euler fused), and a direct module call with device tensors
goes through `engine.decoder_forward`.
"""
import torch
from torch import nn
from torch.nn import functional as F
from torch.utils.checkpoint import checkpoint

from .. import _backend
from .. import _torch_ops as ops
from .embeddings import PatchEmbed
from .normalization import AdaLayerNormZero
from .vision_transformer import use_hip


def _gather_rows(t, ids):
    return t.gather(1, ids.expand(-1, -1, t.size(-1)))


class Projector(nn.Module):
    """fc2(SiLU(fc1(x)))."""

    def __init__(self, dim, mlp_dim=None, out_dim=None):
        super().__init__()
        hidden = mlp_dim or dim
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, out_dim or dim)
        self.activation = nn.SiLU()

    def forward(self, x) -> torch.Tensor:
        return self.fc2(_backend.train_activation(self.fc1(x), 2, F.silu))


class TimeCondEmbed(nn.Module):
    """z' = condition_proj(z) + timestep_proj([cos, sin](t * f)): one vector per token carrying both conditions."""

    def __init__(self, cond_dim, embed_dim, freq_dim=256):
        super().__init__()
        self.freq_dim = freq_dim
        self.time_freq = None  # [1, freq_dim / 2], built on first use
        self.timestep_proj = Projector(freq_dim, embed_dim, embed_dim)
        self.condition_proj = Projector(cond_dim, embed_dim, embed_dim)

    def get_freq_embed(self, timestep, dtype) -> torch.Tensor:
        if self.time_freq is None or self.time_freq.device != timestep.device:
            self.time_freq = ops.timestep_frequencies(self.freq_dim // 2, device=timestep.device)
        feats = ops.sincos_features(timestep.unsqueeze(-1).float(), self.time_freq, sin_first=False)
        return feats.to(dtype=dtype)

    def forward(self, timestep, z) -> torch.Tensor:
        t = self.timestep_proj(self.get_freq_embed(timestep, z.dtype))
        if t.dim() == 2:  # one timestep per sequence: broadcast over its tokens
            t = t.unsqueeze(1)
        return self.condition_proj(z) + t


class DiffusionBlock(nn.Module):
    """AdaLN-Zero residual block: x + gate * LN(proj(LN0(x) (1 + scale) + shift))."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.mlp_checkpointing = False
        self.norm1 = AdaLayerNormZero(dim, num_stats=3, eps=1e-6)
        self.proj = Projector(dim, dim, dim)
        self.norm2 = nn.LayerNorm(dim)

    def forward(self, x, z) -> torch.Tensor:
        recompute = self.mlp_checkpointing and x.requires_grad
        if recompute:
            h, (gate,) = checkpoint(self.norm1, x, z, use_reentrant=False)
            h = checkpoint(self.proj, h, use_reentrant=False)
        else:
            h, (gate,) = self.norm1(x, z)
            h = self.proj(h)
        if _backend.train_norm_supported(h, gamma=self.norm2.weight, gate=gate, res=x):
            # training on the GPU: norm2(h) * gate + x as one HIP row kernel each way (csrc/rownorm_bwd.hip)
            return _backend.autograd().fused_norm(h, gamma=self.norm2.weight, beta=self.norm2.bias, gate=gate, res=x, eps=self.norm2.eps)
        return self.norm2(h) * gate + x


class DiffusionMLP(nn.Module):
    """Noise / velocity predictor applied to single tokens, conditioned on the encoder output of that token."""

    def __init__(self, depth, embed_dim, cond_dim, patch_size=2, image_dim=4):
        super().__init__()
        self.patch_embed = PatchEmbed(image_dim, embed_dim, patch_size)
        self.time_cond_embed = TimeCondEmbed(cond_dim, embed_dim)
        self.blocks = nn.ModuleList([DiffusionBlock(embed_dim) for _ in range(depth)])
        self.norm = AdaLayerNormZero(embed_dim, num_stats=2, eps=1e-6)
        self.head = nn.Linear(embed_dim, patch_size * patch_size * image_dim)

    def forward(self, x, timestep, z, pred_ids=None) -> torch.Tensor:
        """x [S, C, H, W] noisy canvas, z [S, N, D] condition. With `pred_ids` [S, n, 1] only those rows are computed
        and the other rows of the result echo patchify(x)."""
        if use_hip(z):
            return _backend.engine().decoder_forward(self, x, timestep, z, pred_ids)
        tokens = self.patch_embed(x)
        subset = pred_ids is not None
        if subset:
            canvas = self.patch_embed.patchify(x)
            tokens, z = _gather_rows(tokens, pred_ids), _gather_rows(z, pred_ids)
        z = self.time_cond_embed(timestep, z)
        for block in self.blocks:
            tokens = block(tokens, z)
        pred = self.head(self.norm(tokens, z)[0])
        if not subset:
            return pred
        return canvas.scatter(1, pred_ids.expand(-1, -1, pred.size(-1)), pred)
