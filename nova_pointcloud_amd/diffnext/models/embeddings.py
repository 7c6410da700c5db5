"""Embedding layers of the NOVA generator: the module surface of reference diffnext/models/embeddings.py.

Class names, constructor arguments, parameter / buffer names and call semantics follow the reference
(RotaryEmbed3D :27-67, PosEmbed :70-91, VideoPosEmbed :94-115, MotionEmbed :118-136, PatchEmbed :139-166,
TextEmbed :169-206, LabelEmbed :209-223, MaskEmbed :226-286) so that checkpoints and calling code carry over.
The arithmetic lives in `diffnext/_torch_ops.py`; on an MI355X the engine replaces it with `nova_rope_table`,
`nova_embed_canvas`, `nova_build_sequence` and friends (include/nova_hip.h).
"""
from typing import List, Tuple, Union

import numpy as np
import torch
from torch import nn

from .. import _torch_ops as ops


def rotary_axis_dims(dim):
    """Channels given to the (t, h, w) axes (reference :48)."""
    return ops.rotary_channel_split(dim)


# =============================================================================================
# tokens <-> latent images
# =============================================================================================
class PatchEmbed(nn.Module):
    """Non-overlapping patch projection. Remembers the token grid of its last image for patchify / unpatchify."""

    def __init__(self, image_dim, embed_dim, patch_size):
        super().__init__()
        self.image_dim = image_dim
        self.patch_size = patch_size
        self.proj = nn.Conv2d(image_dim, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.height = None
        self.width = None

    @property
    def hw(self) -> Tuple[int, int]:
        return self.height, self.width

    def patchify(self, x) -> torch.Tensor:
        return ops.image_to_patch_rows(x, self.image_dim, self.height, self.width, self.patch_size)

    def unpatchify(self, x) -> torch.Tensor:
        return ops.patch_rows_to_image(x, self.image_dim, self.height, self.width, self.patch_size)

    def forward(self, x) -> torch.Tensor:
        if x.dim() == 3:
            return x  # tokens already
        video = x.dim() == 5
        lead = (x.size(0), x.size(2)) if video else None
        if video:
            x = x.transpose(1, 2).flatten(0, 1)
        self.height = x.size(-2) // self.patch_size
        self.width = x.size(-1) // self.patch_size
        tokens = self.proj(x).flatten(2).transpose(1, 2)
        return tokens.view(lead + tokens.shape[1:]) if video else tokens


class MaskEmbed(nn.Module):
    """Mask-token blending and the random generation order of the masked-autoregressive loop.

    RNG contract (reference :262-270): the FIRST `get_pred_mask` after `pred_ids` was reset draws one uniform
    [B, N, 1] tensor from `self.generator`; its argsort is the generation order of the whole sample.
    """

    def __init__(self, embed_dim, mask_ratios=(0.7, 1.0)):
        super().__init__()
        ratios = list(mask_ratios)
        self.mask_ratios = ratios + [0.25] if len(ratios) == 2 else ratios
        self.bos_token = nn.Parameter(torch.zeros(1, embed_dim))
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))
        nn.init.normal_(self.bos_token, std=0.02)
        nn.init.normal_(self.mask_token, std=0.02)
        self.mask = None
        self.attn_mask = None
        self.generator = None
        self.pred_ids = None
        self.pred_pos = 0
        self.batch_shard = None  # (lo, hi, total): this process generates rows [lo, hi) of a batch of `total` (sharding.py)

    # ---- generation state ---------------------------------------------------------------
    def apply_mask(self, x) -> torch.Tensor:
        return ops.blend_mask_token(x, self.mask, self.mask_token)

    def get_pred_mask(self, num_preds) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.pred_ids is None:
            if self.batch_shard is None:
                draw = torch.empty_like(self.mask).uniform_(generator=self.generator)
            else:  # sharded batch: draw for the global batch, keep this shard's rows (same order as the unsharded run)
                lo, hi, total = self.batch_shard
                draw = self.mask.new_empty((total,) + tuple(self.mask.shape[1:])).uniform_(generator=self.generator)[lo:hi]
            self.pred_ids = draw.argsort(dim=1)
        chosen = self.pred_ids[:, self.pred_pos : self.pred_pos + num_preds]
        self.pred_pos += num_preds
        pred_mask = torch.zeros_like(self.mask).scatter_(1, chosen, 1)
        self.mask = self.mask * (1 - pred_mask)
        return pred_mask, chosen

    # ---- training: frame-level causal attention over [condition ; frame 0 ; frame 1 ; ...] ---
    def get_attn_lens(self, x: Union[torch.Tensor, Tuple[torch.Tensor]], c: torch.Tensor = None) -> List[int]:
        multi = isinstance(x, (tuple, list))
        lens = [t.shape[1:3].numel() for t in x] if multi else [x.size(2)] * x.size(1)
        if c is not None:
            lens[0] += c.size(1)
        return lens

    def get_attn_mask(self, x, c: torch.Tensor = None, persistent=True) -> torch.Tensor:
        if persistent and self.attn_mask is not None:
            return self.attn_mask
        if isinstance(x, (tuple, list)):
            frame = torch.cat([torch.full(t.shape[1:3], i) for i, t in enumerate(x)]).flatten()
        else:
            frame = torch.arange(x.size(1)).repeat_interleave(x.size(2))
        if c is not None:
            frame = torch.cat([frame.new_zeros(c.size(1)), frame])
        mask = ops.frame_causal_mask(frame)
        self.attn_mask = mask.to(device=self.bos_token.device, dtype=self.bos_token.dtype)
        return self.attn_mask

    def forward(self, x) -> torch.Tensor:
        if not self.training:
            if self.mask is None:  # start of a sample: everything masked
                self.mask = x.new_ones(x.shape[:-1] + (1,))
                self.pred_pos = 0
            return self.apply_mask(x)
        # training: keep a random (1 - ratio) share of the tokens, ratio ~ truncated normal around 1
        import scipy.stats as stats

        order = torch.rand(x.shape[:-1] + (1,), device=x.device).argsort(1)
        width = self.mask_ratios[2]
        lo, hi = ((v - 1) / width for v in self.mask_ratios[:2])
        ratio = stats.truncnorm(lo, hi, loc=1, scale=width).rvs(1)[0]
        n_keep = int(np.round((1 - ratio) * order.size(1)))
        prev_ids = order[:, :n_keep]
        self.mask = x.new_ones(order.shape).scatter_(1, prev_ids, 0)
        return self.apply_mask(x), prev_ids


# =============================================================================================
# positions
# =============================================================================================
class RotaryEmbed3D(nn.Identity):
    """3-D rotary embedding over integer (t, h, w) grid positions; no parameters, three non-persistent buffers."""

    class ApplyFunc(object):
        """Callable holding a [S|1, 1, L, d/2, 2, 2] rotation table (`weight`) for q / k [S, heads, L, d]."""

        def __init__(self, weight: torch.Tensor):
            self.weight = weight

        def __call__(self, x: torch.Tensor) -> torch.Tensor:
            self.weight = self.weight.to(dtype=x.dtype)
            return ops.rotate_channel_pairs(self.weight, x)

    def __init__(self, dim=64, base_size=(16, 16), theta=10000.0):
        super().__init__()
        self.dim = dim
        self.base_size = base_size
        self.theta = theta
        for axis, n in enumerate(ops.rotary_channel_split(dim)):
            self.register_buffer(f"scale{axis}", ops.rotary_exponents(n), persistent=False)

    def _axis_inv_freq(self):
        return [torch.pow(self.theta, getattr(self, f"scale{a}").float()).reciprocal() for a in range(3)]

    def inv_freq(self) -> torch.Tensor:
        """1 / theta^scale for all channel pairs, axis order t, h, w: the vector `nova_rope_table` consumes."""
        return torch.cat(self._axis_inv_freq())

    def get_pos(self, t=1, bs=1, hw=None) -> torch.Tensor:
        sizes = [t] + list(self.base_size if hw is None else hw)
        return ops.integer_grid(sizes, device=self.scale1.device, batch=bs)

    def get_func(self, pos: torch.Tensor, pad=0, ids: torch.Tensor = None) -> ApplyFunc:
        return self.ApplyFunc(ops.rotation_table(pos, self._axis_inv_freq(), pad=pad, ids=ids))


class PosEmbed(nn.Module):
    """Fixed 2-D sin-cos table added IN PLACE to the tokens (abs-PE checkpoints)."""

    def __init__(self, dim, base_size=(16, 16)):
        super().__init__()
        self.base_h, self.base_w = base_size
        self.space_embed = None
        self.freq_hw = ops.inverse_frequencies(dim // 4)

    def get_space_embed(self, device=None, dtype=None) -> torch.Tensor:
        n_tokens = self.base_h * self.base_w
        if self.space_embed is None or self.space_embed.size(0) != n_tokens:
            table = ops.sincos_grid_table(self.base_h, self.base_w, self.base_h, self.base_w, self.freq_hw)
            self.space_embed = table.to(device=device, dtype=dtype)
        return self.space_embed

    def forward(self, x) -> torch.Tensor:
        return x.add_(self.get_space_embed(x.device, x.dtype))


class VideoPosEmbed(PosEmbed):
    """PosEmbed plus a learned projection of a sinusoidal frame index (`time_proj`, `norm`)."""

    def __init__(self, dim, base_size):
        super().__init__(dim, base_size=base_size[1:])
        self.base_t = base_size[0]
        self.time_embed = None
        self.norm = nn.LayerNorm(dim)
        self.time_proj = nn.Sequential(nn.Linear(256, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.freq_t = ops.inverse_frequencies(128).unsqueeze(0)

    def get_time_embed(self, t) -> torch.Tensor:
        if self.time_embed is None or self.time_embed.size(0) != t:
            first = self.time_proj[0].weight
            frame = (torch.arange(t, dtype=torch.float32) / (t / self.base_t)).view(-1, 1, 1)
            self.time_embed = ops.sincos_features(frame, self.freq_t).to(device=first.device, dtype=first.dtype)
        return self.norm(self.time_proj(self.time_embed))

    def forward(self, x) -> torch.Tensor:
        if x.dim() == 4:
            x = x.add_(self.get_time_embed(x.size(-3)))
        return super().forward(x)


class MotionEmbed(nn.Module):
    """Motion-flow / fps conditioning tokens (text-to-video only; not on the point-set path)."""

    def __init__(self, dim, base_flow=5, base_fps=12):
        super().__init__()
        self.base_flow = base_flow
        self.base_fps = base_fps
        self.flow_proj = nn.Sequential(nn.Linear(256, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.fps_proj = nn.Sequential(nn.Linear(256, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.freq_m = ops.inverse_frequencies(128).unsqueeze(0)

    def get_embed(self, c, x, k) -> torch.Tensor:
        values = [getattr(self, "base_" + k)] * c.size(0) if x is None else x
        feats = ops.sincos_features(torch.as_tensor(values).view(-1, 1, 1).float(), self.freq_m)
        return getattr(self, k + "_proj")(feats.to(device=c.device, dtype=c.dtype))

    def forward(self, c, flow=None, fps=None) -> torch.Tensor:
        return torch.cat([self.get_embed(c, flow, "flow"), self.get_embed(c, fps, "fps")], dim=1)


# =============================================================================================
# conditions
# =============================================================================================
class TextEmbed(nn.Module):
    """Pads prompt embeddings to a fixed token count with rows of `weight` and projects them to the model width.

    `weight` [512, token_dim] is a persistent buffer; its first rows are also the unconditional prompt.
    """

    def __init__(self, token_dim, embed_dim, num_tokens=256, dropout=0.1):
        super().__init__()
        self.token_dim = token_dim
        self.num_tokens = num_tokens
        self.dropout = dropout
        self.encoders = []
        self.proj = nn.Linear(token_dim, embed_dim)
        self.norm = nn.LayerNorm(embed_dim)
        self.register_buffer("weight", torch.zeros(512, token_dim))
        nn.init.normal_(self.weight, std=0.02)

    def _keep(self, _i=None) -> bool:
        """Prompt dropout of the training recipe: one np.random draw per prompt while training."""
        return not (self.training and self.dropout > 0 and np.random.rand() < self.dropout)

    @torch.no_grad()
    def encode_prompts(self, prompts) -> torch.Tensor:
        pad_rows = self.weight[: self.num_tokens]
        if not isinstance(prompts[0], str):  # precomputed embeddings [len_i, token_dim]
            return ops.pad_prompt_rows(pad_rows, prompts, self._keep)
        tokenizer, encoder = self.encoders
        limit = dict(max_length=self.num_tokens, truncation=True)
        padded_ids = [tokenizer(p, padding="max_length", **limit).input_ids for p in prompts]
        true_len = [len(tokenizer(p, **limit).input_ids) for p in prompts]
        hidden = encoder(torch.as_tensor(padded_ids, device=encoder.device)).last_hidden_state.to(dtype=self.weight.dtype)
        rows = [hidden[i, :n] for i, n in enumerate(true_len)]
        return ops.pad_prompt_rows(pad_rows.to(device=encoder.device), rows, self._keep)

    def forward(self, x) -> torch.Tensor:
        if isinstance(x, (tuple, list)):
            x = self.encode_prompts(x)
        if x.is_cuda and not torch.is_grad_enabled():  # MI355X: projection GEMM + LayerNorm row kernel
            from .. import _backend

            hip = _backend.hip()
            f32 = lambda t: t.detach().float().contiguous()
            rows = x.reshape(-1, x.size(-1)).contiguous()
            h = hip.gemm_bias_act(rows, self.proj.weight.detach().to(x.dtype).contiguous(), f32(self.proj.bias))
            h = hip.row_norm(h, gamma=f32(self.norm.weight), beta=f32(self.norm.bias), eps=self.norm.eps)
            return h.view(*x.shape[:-1], -1)
        return self.norm(self.proj(x))


class LabelEmbed(nn.Module):
    """Class-label embedding table with label dropout (class-to-image only; not on the point-set path)."""

    def __init__(self, embed_dim, num_classes=1000, dropout=0.1):
        super().__init__()
        self.num_classes = num_classes
        self.dropout = dropout
        rows = num_classes + (1 if dropout > 0 else 0)  # the extra row is the "no label" class
        self.weight = nn.Parameter(torch.zeros(rows, embed_dim))
        nn.init.normal_(self.weight, std=0.02)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, input_ids):
        if input_ids.dim() == 1:
            input_ids = input_ids.unsqueeze(-1)
        if self.training and self.dropout > 0:
            kept = torch.rand(input_ids.size(), device=input_ids.device).gt(self.dropout)
            input_ids = input_ids.where(kept, self.num_classes)
        return self.norm(self.weight[input_ids])
