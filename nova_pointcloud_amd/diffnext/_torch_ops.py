"""Functional PyTorch definitions of the arithmetic behind this package's module classes.

The classes in `models/`, `schedulers/` keep the reference's names, constructor signatures and state_dict keys
(they are the drop-in surface); what they compute is written once here, as plain functions on tensors, in the
vocabulary of the HIP kernels that replace them on an MI355X (`include/nova_hip.h`): rotation tables, sin-cos
tables, token plumbing, AdaLN modulation, guidance mixing, the flow-matching grid. Used on CPU tensors and
whenever autograd is on; never by the HIP generation path.
"""
import math

import numpy as np
import torch
from torch.nn import functional as F


# ---------------------------------------------------------------------------------------------
# 3-D rotary positions (what nova_rope_table builds on the device)
# ---------------------------------------------------------------------------------------------
def rotary_channel_split(head_dim):
    """Channels owned by the (frame, row, column) axes: an eighth for time, the rest halved."""
    t = head_dim // 8
    rest = (head_dim - t) // 2
    return [t, rest, rest]


def rotary_exponents(n_channels):
    """Exponent of theta for each channel PAIR of an axis: 0, 2/n, 4/n, ..."""
    return torch.arange(0, n_channels, 2).float().div_(n_channels)


def integer_grid(sizes, device=None, batch=1):
    """All integer coordinates of a box, last axis fastest: [batch, prod(sizes), len(sizes)] f32."""
    axes = [torch.arange(n, device=device, dtype=torch.float32) for n in sizes]
    mesh = torch.meshgrid(*axes, indexing="ij")
    return torch.stack(mesh, dim=-1).view(1, -1, len(sizes)).expand(batch, -1, -1)


def rotation_table(pos, inv_freq_per_axis, pad=0, ids=None):
    """2x2 rotation per (token, channel pair): [bs, 1, pad + n, pairs, 2, 2].

    pos [bs, n_pos, 3]; `ids` [bs, n, 1->3] selects positions; `pad` leading tokens sit at the origin (identity).
    """
    if ids is not None:
        pos = pos.gather(1, ids)
    if pad:
        pos = F.pad(pos, (0, 0, pad, 0), value=0)
    per_axis = []
    for axis, inv in enumerate(inv_freq_per_axis):
        angle = pos[..., axis : axis + 1] * inv.unsqueeze(0)
        cos, sin = angle.cos(), angle.sin()
        per_axis.append(torch.stack([cos, -sin, sin, cos], dim=-1).unflatten(-1, (2, 2)))
    return torch.cat(per_axis, dim=-3).unsqueeze(1)


def rotate_channel_pairs(table, x):
    """Apply a rotation table to adjacent channel pairs of x [S, heads, L, d]."""
    pairs = x.unflatten(-1, (-1, 1, 2))
    return (table[..., 0] * pairs[..., 0] + table[..., 1] * pairs[..., 1]).flatten(3)


# ---------------------------------------------------------------------------------------------
# sin-cos tables (absolute position embeddings, frame / motion / timestep features)
# ---------------------------------------------------------------------------------------------
def inverse_frequencies(n, base=10000.0):
    return 1 / (base ** (torch.arange(n, dtype=torch.float32) / n))


def sincos_features(values, freqs, sin_first=True):
    """[..., 2 * len(freqs)]: sin and cos of value * freq, concatenated in the requested order."""
    angle = values * freqs
    parts = [angle.sin(), angle.cos()] if sin_first else [angle.cos(), angle.sin()]
    return torch.cat(parts, dim=-1)


def sincos_grid_table(grid_h, grid_w, base_h, base_w, freqs):
    """[grid_h * grid_w, 4 * len(freqs)] table, x features first, row-major tokens."""
    ys = torch.arange(grid_h, dtype=torch.float32) * (base_h / grid_h)
    xs = torch.arange(grid_w, dtype=torch.float32) * (base_w / grid_w)
    gx, gy = torch.meshgrid(xs, ys, indexing="xy")
    fx = gx.reshape(-1, 1) * freqs.unsqueeze(0)
    fy = gy.reshape(-1, 1) * freqs.unsqueeze(0)
    return torch.cat([fx.sin(), fx.cos(), fy.sin(), fy.cos()], dim=-1)


def timestep_frequencies(n_pairs, device=None):
    """exp(-ln(1e4) k / n) for k < n, as a [1, n] row."""
    k = torch.arange(n_pairs, dtype=torch.float32, device=device)
    return k.mul(-math.log(10000.0) / n_pairs).exp().unsqueeze(0)


# ---------------------------------------------------------------------------------------------
# token plumbing
# ---------------------------------------------------------------------------------------------
def image_to_patch_rows(x, channels, grid_h, grid_w, patch):
    """[B, C, H, W] -> [B, N, patch*patch*C]; a patch vector is ordered (row, column, channel)."""
    x = x.reshape(-1, channels, grid_h, patch, grid_w, patch)
    return x.permute(0, 2, 4, 3, 5, 1).reshape(x.size(0), grid_h * grid_w, patch * patch * channels)


def patch_rows_to_image(x, channels, grid_h, grid_w, patch):
    x = x.reshape(-1, grid_h, grid_w, patch, patch, channels)
    return x.permute(0, 5, 1, 3, 2, 4).reshape(x.size(0), channels, grid_h * patch, grid_w * patch)


def blend_mask_token(x, mask, token):
    """Known rows keep x, masked rows (mask = 1) become the mask token."""
    return x * (1 - mask) + token * mask


def frame_causal_mask(frame_of_token):
    """Additive attention mask: a token sees the tokens of its own and earlier frames."""
    visible = frame_of_token.unsqueeze(1) >= frame_of_token.unsqueeze(0)
    return torch.zeros(visible.shape).masked_fill_(~visible, -float("inf"))


def pad_prompt_rows(padding_rows, prompts, keep):
    """[B, T, dim]: every prompt written over the first rows of a copy of the padding table (if `keep(i)`)."""
    out = padding_rows.expand(len(prompts), -1, -1).clone()
    for i, p in enumerate(prompts):
        if keep(i):
            out[i, : p.shape[0]] = torch.as_tensor(p, device=out.device).to(out.dtype)
    return out


# ---------------------------------------------------------------------------------------------
# AdaLN, guidance, flow matching
# ---------------------------------------------------------------------------------------------
def adaln_modulate(normed, scale, shift):
    return normed * (1 + scale) + shift


def guided(cond, uncond, weight):
    return uncond + (cond - uncond) * weight


def clamp_to_cond_norm(x, cond, floor):
    """Rescale x towards the per-sample norm of `cond`, the factor clamped to [floor, 1]."""
    dims = tuple(range(1, x.dim()))
    factor = cond.norm(dim=dims, keepdim=True) / x.norm(dim=dims, keepdim=True)
    return x * factor.clamp(floor, 1)


def shift_sigmas(sigmas, shift):
    return shift * sigmas / (1 + (shift - 1) * sigmas)


def training_sigma_grid(num_train_timesteps, shift, dynamic):
    """Descending sigmas 1 ... 1/T (float32 numpy), shifted unless the shift is dynamic."""
    s = np.arange(1, num_train_timesteps + 1, dtype="float32")[::-1] / num_train_timesteps
    return s if dynamic else shift_sigmas(s, shift)


def sampling_sigma_grid(sigma_max, sigma_min, num_train_timesteps, steps):
    t = np.linspace(sigma_max * num_train_timesteps, sigma_min * num_train_timesteps, steps, dtype="float32")
    return t / num_train_timesteps
