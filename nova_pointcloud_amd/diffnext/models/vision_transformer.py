"""ViT encoder of the NOVA generator (reference diffnext/models/vision_transformer.py).

Same classes / ctor signatures / state_dict keys as the reference: MLP :28-38, Attention :41-64,
Block :67-92 (POST-norm residual), VisionTransformer :95-146 (MAE-style split: the first
`encoder_depth` blocks see [condition ; known tokens], the rest [condition ; full canvas]).

Execution: with CUDA (ROCm) tensors and autograd off, `Block.forward` runs on libnova_hip.so
(`nova_vit_blocks_forward`: fused QKV+RoPE GEMM, flash attention reading q/k/v in place,
GEMM+GELU, fused LayerNorm+residual) and raises if the library is missing; the full generation
loop bypasses even this and drives whole block stacks from nova_pointcloud_amd/engine.py.
With CPU tensors or autograd on (training), the PyTorch definition below runs.
"""
from typing import Tuple

import torch
from torch import nn
from torch.nn import functional as F
from torch.utils.checkpoint import checkpoint

from .. import _backend
from .embeddings import PatchEmbed, RotaryEmbed3D


def use_hip(x: torch.Tensor) -> bool:
    """The one dispatch rule of this package: device tensors without autograd go to the HIP kernels."""
    return x.is_cuda and not torch.is_grad_enabled()


class MLP(nn.Module):
    def __init__(self, dim, mlp_ratio=4):
        super().__init__()
        self.fc1 = nn.Linear(dim, int(dim * mlp_ratio))
        self.fc2 = nn.Linear(int(dim * mlp_ratio), dim)
        self.activation = nn.GELU()

    def forward(self, x) -> torch.Tensor:
        if use_hip(x):
            hip = _backend.hip()
            h = hip.gemm_bias_act(x.reshape(-1, x.size(-1)).contiguous(), _w(self.fc1, x), _b(self.fc1), hip.ACT_GELU_ERF)
            return hip.gemm_bias_act(h, _w(self.fc2, x), _b(self.fc2)).view(*x.shape[:-1], -1)
        return self.fc2(self.activation(self.fc1(x)))


def _w(linear, x):
    w = linear.weight.detach()
    return w if (w.dtype == x.dtype and w.is_contiguous()) else w.to(x.dtype).contiguous()


def _b(linear):
    return None if linear.bias is None else linear.bias.detach().float().contiguous()


def rope_table_from_func(pe_func, S):
    """[S|1, 1, L, d/2, 2, 2] rotation matrices -> the (cos, sin) table [nb, L, d/2, 2] f32 of libnova_hip."""
    w = pe_func.weight[:, 0]
    return torch.stack([w[..., 0, 0], w[..., 1, 0]], dim=-1).float().contiguous()


class Attention(nn.Module):
    """Multi-head self-attention with a fused QKV projection and optional RoPE / KV cache."""

    def __init__(self, dim, num_heads, qkv_bias=True):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.attn_mask, self.cache_kv, self.pe_func, self.flex_attn = None, None, None, None

    def forward(self, x) -> torch.Tensor:
        S, L, D = x.shape
        if use_hip(x) and self.attn_mask is None and not self.cache_kv:
            hip = _backend.hip()
            rope = rope_table_from_func(self.pe_func, S) if self.pe_func else None
            qkv = hip.qkv_rope(x.reshape(S * L, D).contiguous(), _w(self.qkv, x), _b(self.qkv), rope, S, L, self.num_heads)
            o = hip.attn_fwd_packed(qkv, S, L, self.num_heads)
            return hip.gemm_bias_act(o, _w(self.proj, x), _b(self.proj)).view(S, L, D)
        qkv = self.qkv(x).view(S, L, 3, self.num_heads, self.head_dim)
        q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)
        if self.pe_func:
            q, k = self.pe_func(q), self.pe_func(k)
        if self.cache_kv:  # frames of a video share keys/values of earlier frames (T > 1)
            if isinstance(self.cache_kv, list):
                k = self.cache_kv[0] = torch.cat([self.cache_kv[0], k], dim=2)
                v = self.cache_kv[1] = torch.cat([self.cache_kv[1], v], dim=2)
            else:
                self.cache_kv = [k, v]
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=self.attn_mask)
        return self.proj(o.transpose(1, 2).flatten(2))


class Block(nn.Module):
    """x <- LN1(attn(x)) + x ; x <- LN2(mlp(x)) + x."""

    def __init__(self, dim, num_heads, mlp_ratio=4, qkv_bias=True):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = Attention(dim, num_heads, qkv_bias=qkv_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = MLP(dim, mlp_ratio=mlp_ratio)
        self.attn_checkpointing, self.mlp_checkpointing = False, False

    def forward_attn(self, x) -> torch.Tensor:
        return self.norm1(self.attn(x))

    def forward_mlp(self, x) -> torch.Tensor:
        return self.norm2(self.mlp(x))

    def forward_ckpt(self, x, name) -> torch.Tensor:
        fn = getattr(self, f"forward_{name}")
        if getattr(self, f"{name}_checkpointing", False) and x.requires_grad:
            return checkpoint(fn, x, use_reentrant=False)
        return fn(x)

    def forward(self, x, pe_func: callable = None) -> torch.Tensor:
        self.attn.pe_func = pe_func
        if use_hip(x) and self.attn.attn_mask is None and not self.attn.cache_kv:
            return _backend.engine().block_stack_forward([self], x, pe_func)
        x = self.forward_ckpt(x, "attn") + x
        return self.forward_ckpt(x, "mlp") + x


class VisionTransformer(nn.Module):
    def __init__(self, depth, embed_dim, num_heads, mlp_ratio=4, patch_size=2, image_size=32, image_dim=4,
                 encoder_depth=None):
        super().__init__()
        self.embed_dim, self.image_size, self.image_dim = embed_dim, image_size, image_dim
        self.patch_embed = PatchEmbed(image_dim, embed_dim, patch_size)
        self.pos_embed, self.rope = nn.Identity(), RotaryEmbed3D(embed_dim // num_heads)
        self.blocks = nn.ModuleList(Block(embed_dim, num_heads, mlp_ratio) for _ in range(depth))
        self.norm, self.mixer = nn.LayerNorm(embed_dim), nn.Identity()
        self.encoder_depth = len(self.blocks) // 2 if encoder_depth is None else encoder_depth
        self.flex_attn = None  # block-causal FlexAttention of the reference is dead code at inference

    def prepare_pe(self, c=None, ids=None, pos=None) -> Tuple[callable, callable]:
        pad = 0 if c is None else c.size(1)
        full = self.rope.get_func(pos, pad)
        known = self.rope.get_func(pos, pad, ids.expand(-1, -1, 3)) if ids is not None else full
        return known, full

    def enable_kvcache(self, mode=True):
        for blk in self.blocks:
            blk.attn.cache_kv = mode

    def _run(self, blocks, x, pe):
        if use_hip(x) and len(blocks) and blocks[0].attn.attn_mask is None and not blocks[0].attn.cache_kv:
            return _backend.engine().block_stack_forward(list(blocks), x, pe)
        for blk in blocks:
            x = blk(x, pe)
        return x

    def forward(self, x, c=None, prev_ids=None, pos=None) -> torch.Tensor:
        x, prev_ids = x if isinstance(x, (tuple, list)) else (x, prev_ids)
        prev_ids = prev_ids if self.encoder_depth else None
        x = canvas = self.pos_embed(self.patch_embed(x))
        pe_known, pe_full = self.prepare_pe(c, prev_ids, pos) if pos is not None else (None, None)
        n_cond = 0 if c is None else c.size(1)
        if prev_ids is not None:  # keep only the already generated tokens, in generation order
            prev_ids = prev_ids.expand(-1, -1, x.size(-1))
            x = x.gather(1, prev_ids)
        x = x if c is None else torch.cat([c, x], dim=1)
        x = self._run(self.blocks[: self.encoder_depth], x, pe_known)
        if prev_ids is not None:  # put them back on the mask-token canvas, keep the updated condition
            tokens = canvas.to(dtype=x.dtype).scatter(1, prev_ids, x[:, n_cond:])
            x = torch.cat([x[:, :n_cond], tokens], dim=1) if n_cond else tokens
        x = self._run(self.blocks[self.encoder_depth :], x, pe_full)
        x = x[:, n_cond:] if n_cond else x
        if use_hip(x):
            hip = _backend.hip()
            flat = x.reshape(-1, x.size(-1)).contiguous()
            out = hip.row_norm(flat, gamma=self.norm.weight.detach().float().contiguous(),
                               beta=self.norm.bias.detach().float().contiguous(), eps=self.norm.eps)
            return out.view(x.shape)
        return self.norm(x)
