"""Embedding layers of the NOVA generator (reference diffnext/models/embeddings.py).

Mirrors class names, constructor signatures, parameter/buffer names and call semantics of
RotaryEmbed3D :27-67, PosEmbed :70-91, VideoPosEmbed :94-115, MotionEmbed :118-136,
PatchEmbed :139-166, TextEmbed :169-206, LabelEmbed :209-223, MaskEmbed :226-286.
These are the PyTorch definitions (CPU / autograd). On an MI355X the engine replaces them with
`nova_rope_table`, `nova_embed_canvas`, `nova_build_sequence` (see include/nova_hip.h).
"""
from typing import List, Tuple, Union

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F


def rotary_axis_dims(dim):
    """Channels given to the (t, h, w) axes: [d/8, rest/2, rest/2] (reference :48)."""
    return [dim // 8] + [(dim - dim // 8) // 2] * 2


class RotaryEmbed3D(nn.Identity):
    """3-D rotary embedding over (t, h, w) integer grid positions."""

    class ApplyFunc(object):
        """Rotates adjacent channel pairs of q / k [S, heads, L, d] by a [S|1, 1, L, d/2, 2, 2] table."""

        def __init__(self, weight: torch.Tensor):
            self.weight = weight

        def __call__(self, x: torch.Tensor) -> torch.Tensor:
            w = self.weight = self.weight.to(dtype=x.dtype)
            pairs = x.unflatten(-1, (-1, 1, 2))  # [..., d/2, 1, 2]
            return (w[..., 0] * pairs[..., 0] + w[..., 1] * pairs[..., 1]).flatten(3)

    def __init__(self, dim=64, base_size=(16, 16), theta=10000.0):
        super().__init__()
        self.dim, self.base_size, self.theta = dim, base_size, theta
        for i, rd in enumerate(rotary_axis_dims(dim)):
            self.register_buffer("scale%d" % i, torch.arange(0, rd, 2).float().div_(rd), persistent=False)

    def inv_freq(self) -> torch.Tensor:
        """1 / theta^scale for all pairs, axis order t, h, w: what `nova_rope_table` consumes."""
        return torch.cat([torch.pow(self.theta, getattr(self, "scale%d" % i).float()).reciprocal() for i in range(3)])

    def get_pos(self, t=1, bs=1, hw=None) -> torch.Tensor:
        dev = self.scale1.device
        sizes = [t] + list(hw or self.base_size)
        axes = torch.meshgrid(*[torch.arange(n, device=dev, dtype=torch.float32) for n in sizes], indexing="ij")
        return torch.stack(axes, dim=-1).view(1, -1, 3).expand(bs, -1, -1)

    def get_func(self, pos: torch.Tensor, pad=0, ids: torch.Tensor = None) -> ApplyFunc:
        if ids is not None:
            pos = pos.gather(1, ids)
        if pad:
            pos = F.pad(pos, (0, 0, pad, 0), value=0)
        tables = []
        for i in range(3):
            inv = torch.pow(self.theta, getattr(self, "scale%d" % i).float()).reciprocal()
            ang = pos[..., i : i + 1] * inv.unsqueeze(0)
            rot = torch.stack([ang.cos(), -ang.sin(), ang.sin(), ang.cos()], dim=-1)
            tables.append(rot.unflatten(-1, (2, 2)))
        return self.ApplyFunc(torch.cat(tables, dim=-3).unsqueeze(1))


class PosEmbed(nn.Module):
    """Fixed 2-D sin-cos position table added in place (abs-PE checkpoints)."""

    def __init__(self, dim, base_size=(16, 16)):
        super().__init__()
        (self.base_h, self.base_w), self.space_embed = base_size, None
        self.freq_hw = 1 / (10000 ** (torch.arange(dim // 4, dtype=torch.float32) / (dim // 4)))

    def get_space_embed(self, device=None, dtype=None) -> torch.Tensor:
        h, w = self.base_h, self.base_w
        if self.space_embed is not None and self.space_embed.size(0) == h * w:
            return self.space_embed
        ys = torch.arange(h, dtype=torch.float32) * (self.base_h / h)
        xs = torch.arange(w, dtype=torch.float32) * (self.base_w / w)
        gx, gy = torch.meshgrid(xs, ys, indexing="xy")
        fx, fy = (g.reshape(-1, 1) * self.freq_hw.unsqueeze(0) for g in (gx, gy))
        table = torch.cat([fx.sin(), fx.cos(), fy.sin(), fy.cos()], dim=-1)
        self.space_embed = table.to(device=device, dtype=dtype)
        return self.space_embed

    def forward(self, x) -> torch.Tensor:
        return x.add_(self.get_space_embed(x.device, x.dtype))


class VideoPosEmbed(PosEmbed):
    """PosEmbed + a learned projection of a sinusoidal frame index."""

    def __init__(self, dim, base_size):
        super().__init__(dim, base_size=base_size[1:])
        self.base_t, self.time_embed, self.norm = base_size[0], None, nn.LayerNorm(dim)
        self.time_proj = nn.Sequential(nn.Linear(256, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.freq_t = 1 / (10000 ** (torch.arange(128, dtype=torch.float32).unsqueeze(0) / 128))

    def get_time_embed(self, t) -> torch.Tensor:
        if self.time_embed is None or t != self.time_embed.size(0):
            w0 = self.time_proj[0].weight
            frames = torch.arange(t, dtype=torch.float32) / (t / self.base_t)
            ang = frames.view(-1, 1, 1) * self.freq_t
            self.time_embed = torch.cat([ang.sin(), ang.cos()], dim=-1).to(device=w0.device, dtype=w0.dtype)
        return self.norm(self.time_proj(self.time_embed))

    def forward(self, x) -> torch.Tensor:
        if x.dim() == 4:
            x = x.add_(self.get_time_embed(x.size(-3)))
        return x.add_(self.get_space_embed(x.device, x.dtype))


class MotionEmbed(nn.Module):
    """Motion-flow / fps conditioning tokens (T2V only; not on the point-set path)."""

    def __init__(self, dim, base_flow=5, base_fps=12):
        super().__init__()
        self.base_flow, self.base_fps = base_flow, base_fps
        self.flow_proj = nn.Sequential(nn.Linear(256, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.fps_proj = nn.Sequential(nn.Linear(256, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.freq_m = 1 / (10000 ** (torch.arange(128, dtype=torch.float32).unsqueeze(0) / 128))

    def get_embed(self, c, x, k) -> torch.Tensor:
        x = [getattr(self, f"base_{k}")] * c.size(0) if x is None else x
        ang = torch.as_tensor(x).view(-1, 1, 1).float() * self.freq_m
        feats = torch.cat([ang.sin(), ang.cos()], dim=-1).to(device=c.device, dtype=c.dtype)
        return getattr(self, f"{k}_proj")(feats)

    def forward(self, c, flow=None, fps=None) -> torch.Tensor:
        return torch.cat([self.get_embed(c, flow, "flow"), self.get_embed(c, fps, "fps")], dim=1)


class PatchEmbed(nn.Module):
    """Non-overlapping patch projection; remembers the token grid for patchify / unpatchify."""

    def __init__(self, image_dim, embed_dim, patch_size):
        super().__init__()
        self.height = self.width = None
        self.image_dim, self.patch_size = image_dim, patch_size
        self.proj = nn.Conv2d(image_dim, embed_dim, patch_size, patch_size)

    @property
    def hw(self) -> Tuple[int, int]:
        return self.height, self.width

    def patchify(self, x) -> torch.Tensor:
        """[B, C, H, W] -> [B, N, p*p*C] with the patch vector ordered (row, col, channel)."""
        p = self.patch_size
        x = x.reshape(-1, self.image_dim, self.height, p, self.width, p)
        return x.permute(0, 2, 4, 3, 5, 1).reshape(x.size(0), self.height * self.width, p * p * self.image_dim)

    def unpatchify(self, x) -> torch.Tensor:
        p = self.patch_size
        x = x.reshape(-1, self.height, self.width, p, p, self.image_dim)
        return x.permute(0, 5, 1, 3, 2, 4).reshape(x.size(0), self.image_dim, self.height * p, self.width * p)

    def forward(self, x) -> torch.Tensor:
        if x.dim() == 3:  # already tokens
            return x
        frames = (x.size(0), x.size(2)) if x.dim() == 5 else None
        x = x.transpose(1, 2).flatten(0, 1) if frames else x
        self.height, self.width = x.size(-2) // self.patch_size, x.size(-1) // self.patch_size
        x = self.proj(x).flatten(2).transpose(1, 2)
        return x.view(frames + x.shape[1:]) if frames else x


class TextEmbed(nn.Module):
    """Pads prompt embeddings to a fixed token count and projects them to the model width."""

    def __init__(self, token_dim, embed_dim, num_tokens=256, dropout=0.1):
        super().__init__()
        self.token_dim, self.num_tokens, self.encoders = token_dim, num_tokens, []
        self.proj, self.norm = nn.Linear(token_dim, embed_dim), nn.LayerNorm(embed_dim)
        self.register_buffer("weight", torch.zeros(512, token_dim))  # padding rows; also the null prompt
        nn.init.normal_(self.weight, std=0.02)
        self.dropout = dropout

    def _dropped(self) -> bool:
        return self.training and self.dropout > 0 and np.random.rand() < self.dropout

    @torch.no_grad()
    def encode_prompts(self, prompts) -> torch.Tensor:
        device, dtype = self.weight.device, self.weight.dtype
        x = self.weight[: self.num_tokens].expand(len(prompts), -1, -1).clone()
        if not isinstance(prompts[0], str):  # precomputed embeddings [len_i, token_dim]
            for i, p in enumerate(prompts):
                if not self._dropped():
                    x[i, : p.shape[0]] = torch.as_tensor(p, device=device).to(dtype)
            return x
        tokenizer, encoder = self.encoders
        limit = {"max_length": self.num_tokens, "truncation": True}
        ids = [tokenizer(p, padding="max_length", **limit).input_ids for p in prompts]
        lens = [len(tokenizer(p, **limit).input_ids) for p in prompts]
        embeds = encoder(torch.as_tensor(ids, device=encoder.device)).last_hidden_state.to(dtype=dtype)
        x = x.to(device=encoder.device)
        for i, n in enumerate(lens):
            if not self._dropped():
                x[i, :n] = embeds[i, :n]
        return x

    def forward(self, x) -> torch.Tensor:
        x = self.encode_prompts(x) if isinstance(x, (tuple, list)) else x
        if x.is_cuda and not torch.is_grad_enabled():  # MI355X: projection GEMM + LayerNorm row kernel
            from .. import _backend

            hip = _backend.hip()
            w = self.proj.weight.detach().to(x.dtype).contiguous()
            h = hip.gemm_bias_act(x.reshape(-1, x.size(-1)).contiguous(), w, self.proj.bias.detach().float().contiguous())
            h = hip.row_norm(h, gamma=self.norm.weight.detach().float().contiguous(),
                             beta=self.norm.bias.detach().float().contiguous(), eps=self.norm.eps)
            return h.view(*x.shape[:-1], -1)
        return self.norm(self.proj(x))


class LabelEmbed(nn.Module):
    """Class-label embedding table (C2I; not on the point-set path)."""

    def __init__(self, embed_dim, num_classes=1000, dropout=0.1):
        super().__init__()
        self.dropout, self.num_classes = dropout, num_classes
        self.weight = nn.Parameter(torch.zeros(num_classes + (dropout > 0), embed_dim))
        nn.init.normal_(self.weight, std=0.02)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, input_ids):
        input_ids = input_ids.unsqueeze(-1) if input_ids.dim() == 1 else input_ids
        if self.training and self.dropout > 0:
            keep = torch.rand(input_ids.size(), device=input_ids.device).gt(self.dropout)
            input_ids = input_ids.where(keep, self.num_classes)
        return self.norm(self.weight[input_ids])


class MaskEmbed(nn.Module):
    """Mask-token blending and the random generation order of the masked-AR loop.

    RNG contract (reference :262-270): the FIRST `get_pred_mask` after `pred_ids` was reset draws
    one uniform [B, N, 1] tensor from `self.generator`; its argsort is the generation order.
    """

    def __init__(self, embed_dim, mask_ratios=(0.7, 1.0)):
        super().__init__()
        self.mask_ratios = list(mask_ratios) + ([0.25] if len(mask_ratios) == 2 else [])
        self.bos_token = nn.Parameter(torch.zeros(1, embed_dim))
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))
        for tok in (self.bos_token, self.mask_token):
            nn.init.normal_(tok, std=0.02)
        self.mask, self.attn_mask = None, None
        self.pred_ids, self.pred_pos, self.generator = None, 0, None

    def get_attn_lens(self, x: Union[torch.Tensor, Tuple[torch.Tensor]], c: torch.Tensor = None) -> List[int]:
        if isinstance(x, (tuple, list)):
            lens = [t.shape[1:3].numel() for t in x]
        else:
            lens = [x.size(2)] * x.size(1)
        lens[0] += c.size(1) if c is not None else 0
        return lens

    def get_attn_mask(self, x, c: torch.Tensor = None, persistent=True) -> torch.Tensor:
        """Block-causal (frame level) additive mask used by T > 1 training."""
        if self.attn_mask is not None and persistent:
            return self.attn_mask
        if isinstance(x, (tuple, list)):
            frame = torch.cat([torch.full(t.shape[1:3], i) for i, t in enumerate(x)]).flatten()
        else:
            frame = torch.arange(x.size(1)).repeat_interleave(x.size(2))
        if c is not None:
            frame = torch.cat([torch.zeros(c.size(1), dtype=frame.dtype), frame])
        allowed = frame.unsqueeze(1) >= frame.unsqueeze(0)
        mask = torch.zeros(allowed.shape).masked_fill_(~allowed, -float("inf"))
        self.attn_mask = mask.to(device=self.bos_token.device, dtype=self.bos_token.dtype)
        return self.attn_mask

    def get_pred_mask(self, num_preds) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.pred_ids is None:
            u = torch.empty_like(self.mask).uniform_(generator=self.generator)
            self.pred_ids = u.argsort(dim=1)
        ids = self.pred_ids[:, self.pred_pos : self.pred_pos + num_preds]
        pred_mask = torch.zeros_like(self.mask).scatter_(1, ids, 1)
        self.pred_pos += num_preds
        self.mask = self.mask * (1 - pred_mask)
        return pred_mask, ids

    def apply_mask(self, x) -> torch.Tensor:
        return x * (1 - self.mask) + self.mask_token * self.mask

    def forward(self, x) -> torch.Tensor:
        if self.training:
            import scipy.stats as stats

            u = torch.rand(x.shape[:-1] + (1,), device=x.device)
            lo, hi = [(v - 1) / self.mask_ratios[2] for v in self.mask_ratios[:2]]
            ratio = stats.truncnorm(lo, hi, loc=1, scale=self.mask_ratios[2]).rvs(1)[0]
            prev_ids = u.argsort(1)[:, : int(np.round((1 - ratio) * u.size(1)))]
            self.mask = x.new_ones(u.shape).scatter_(1, prev_ids, 0)
            return self.apply_mask(x), prev_ids
        if self.mask is None:
            self.mask, self.pred_pos = x.new_ones(x.shape[:-1] + (1,)), 0
        return self.apply_mask(x)
