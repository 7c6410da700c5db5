"""ViT encoder of the NOVA generator: the module surface of reference diffnext/models/vision_transformer.py.

Same classes, constructor signatures and state_dict keys as the reference - MLP :28-38, Attention :41-64,
Block :67-92 (POST-norm residual), VisionTransformer :95-146 (MAE-style split: the first `encoder_depth` blocks see
[condition ; known tokens], the rest [condition ; full canvas]).

Execution. With device tensors and autograd off, whole block stacks run on libnova_hip.so
(`nova_vit_blocks_forward`: fused QKV + RoPE GEMM, flash attention reading q / k / v in place, GEMM + GELU, fused
LayerNorm + residual) and raise if the library is missing; the generation loop bypasses even this and drives the
stacks from nova_pointcloud_amd/engine.py. With CPU tensors or autograd on (training) the PyTorch definition runs, with
the attention itself (forward and backward) on the HIP kernels for bf16 device tensors of head_dim 64 / 96.
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


def _w(linear, x):
    """Weight of an nn.Linear in the activation dtype, contiguous (no copy when it already is)."""
    w = linear.weight.detach()
    same = w.dtype == x.dtype and w.is_contiguous()
    return w if same else w.to(x.dtype).contiguous()


def _b(linear):
    return linear.bias.detach().float().contiguous() if linear.bias is not None else None


def _f32(t):
    return t.detach().float().contiguous()


def rope_table_from_func(pe_func, S):
    """[S|1, 1, L, d/2, 2, 2] rotation matrices -> the (cos, sin) table [nb, L, d/2, 2] f32 libnova_hip consumes."""
    rot = pe_func.weight[:, 0]
    return torch.stack([rot[..., 0, 0], rot[..., 1, 0]], dim=-1).float().contiguous()


def _plain_attention(attn) -> bool:
    """No additive mask and no KV cache: the case the HIP block kernels cover."""
    return attn.attn_mask is None and not attn.cache_kv


class Attention(nn.Module):
    """Multi-head self-attention: fused QKV projection, optional RoPE on q / k, optional KV cache (video frames)."""

    def __init__(self, dim, num_heads, qkv_bias=True):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.qkv = nn.Linear(dim, 3 * dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.pe_func = None
        self.attn_mask = None
        self.cache_kv = None
        self.flex_attn = None

    def _extend_cache(self, k, v):
        if not isinstance(self.cache_kv, list):  # first frame: start the cache
            self.cache_kv = [k, v]
            return k, v
        self.cache_kv[0] = torch.cat([self.cache_kv[0], k], dim=2)
        self.cache_kv[1] = torch.cat([self.cache_kv[1], v], dim=2)
        return self.cache_kv[0], self.cache_kv[1]

    def forward(self, x) -> torch.Tensor:
        S, L, D = x.shape
        if use_hip(x) and _plain_attention(self):
            hip = _backend.hip()
            table = rope_table_from_func(self.pe_func, S) if self.pe_func else None
            qkv = hip.qkv_rope(x.reshape(S * L, D).contiguous(), _w(self.qkv, x), _b(self.qkv), table, S, L, self.num_heads)
            merged = hip.attn_fwd_packed(qkv, S, L, self.num_heads)
            return hip.gemm_bias_act(merged, _w(self.proj, x), _b(self.proj)).view(S, L, D)
        q, k, v = self.qkv(x).view(S, L, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4).unbind(0)
        if self.pe_func:
            q = self.pe_func(q)
            k = self.pe_func(k)
        if self.cache_kv:
            k, v = self._extend_cache(k, v)
        if torch.is_grad_enabled() and not self.cache_kv and _backend.train_attention_supported(q, self.attn_mask):
            out = _backend.autograd().attention(q, k, v, self.attn_mask)  # training on the GPU: HIP flash forward + backward (csrc/attn_bwd.hip)
        else:
            out = F.scaled_dot_product_attention(q, k, v, attn_mask=self.attn_mask)
        return self.proj(out.transpose(1, 2).flatten(2))


class MLP(nn.Module):
    """fc2(GELU_erf(fc1(x)))."""

    def __init__(self, dim, mlp_ratio=4):
        super().__init__()
        hidden = int(dim * mlp_ratio)
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)
        self.activation = nn.GELU()

    def forward(self, x) -> torch.Tensor:
        if not use_hip(x):
            return self.fc2(_backend.train_activation(self.fc1(x), 1, self.activation))
        hip = _backend.hip()
        rows = x.reshape(-1, x.size(-1)).contiguous()
        hidden = hip.gemm_bias_act(rows, _w(self.fc1, x), _b(self.fc1), hip.ACT_GELU_ERF)
        return hip.gemm_bias_act(hidden, _w(self.fc2, x), _b(self.fc2)).view(*x.shape[:-1], -1)


class Block(nn.Module):
    """Post-norm residual block: x <- LN1(attn(x)) + x, then x <- LN2(mlp(x)) + x."""

    def __init__(self, dim, num_heads, mlp_ratio=4, qkv_bias=True):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = Attention(dim, num_heads, qkv_bias=qkv_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = MLP(dim, mlp_ratio=mlp_ratio)
        self.attn_checkpointing = False
        self.mlp_checkpointing = False

    def forward_attn(self, x) -> torch.Tensor:
        return self.norm1(self.attn(x))

    def forward_mlp(self, x) -> torch.Tensor:
        return self.norm2(self.mlp(x))

    def forward_ckpt(self, x, name) -> torch.Tensor:
        branch = getattr(self, "forward_" + name)
        recompute = getattr(self, name + "_checkpointing", False) and x.requires_grad
        return checkpoint(branch, x, use_reentrant=False) if recompute else branch(x)

    def forward(self, x, pe_func: callable = None) -> torch.Tensor:
        self.attn.pe_func = pe_func
        if use_hip(x) and _plain_attention(self.attn):
            return _backend.engine().block_stack_forward([self], x, pe_func)
        if _backend.train_norm_supported(x, gamma=self.norm1.weight, res=x):
            # training on the GPU: branch + LayerNorm + residual, the norm and the add as ONE row kernel forward and ONE backward
            # (csrc/rowops.hip, csrc/rownorm_bwd.hip) - the same arithmetic as the two lines below. The check here is on x alone
            # (device, width); whether the kernel applies is decided per branch on the tensors it would actually get
            return self._post_norm_fused(self._post_norm_fused(x, "attn"), "mlp")
        x = x + self.forward_ckpt(x, "attn")
        return x + self.forward_ckpt(x, "mlp")

    def _post_norm_fused(self, x, name):
        inner, norm = (self.attn, self.norm1) if name == "attn" else (self.mlp, self.norm2)
        fused = _backend.autograd().fused_norm

        def branch(t):
            y = inner(t)
            # under torch.autocast (the reference trainer: train_newloss.py:1049, f32 parameters) the branch output is fp16 / bf16
            # while the residual stream stays f32: the row kernel takes ONE storage type, so such a pair goes the torch way
            if _backend.train_norm_supported(y, gamma=norm.weight, res=t):
                return fused(y, gamma=norm.weight, beta=norm.bias, res=t, eps=norm.eps)
            return norm(y) + t

        recompute = getattr(self, name + "_checkpointing", False) and x.requires_grad
        return checkpoint(branch, x, use_reentrant=False) if recompute else branch(x)


class VisionTransformer(nn.Module):
    """Encoder over [condition ; image tokens]; `encoder_depth` blocks run on the known tokens only."""

    def __init__(self, depth, embed_dim, num_heads, mlp_ratio=4, patch_size=2, image_size=32, image_dim=4,
                 encoder_depth=None):
        super().__init__()
        self.embed_dim = embed_dim
        self.image_size = image_size
        self.image_dim = image_dim
        self.patch_embed = PatchEmbed(image_dim, embed_dim, patch_size)
        self.pos_embed = nn.Identity()
        self.rope = RotaryEmbed3D(embed_dim // num_heads)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim)
        self.mixer = nn.Identity()
        self.encoder_depth = depth // 2 if encoder_depth is None else encoder_depth
        self.flex_attn = None  # the reference's block-causal FlexAttention is dead code at inference

    def enable_kvcache(self, mode=True):
        for block in self.blocks:
            block.attn.cache_kv = mode

    def prepare_pe(self, c=None, ids=None, pos=None) -> Tuple[callable, callable]:
        """(rotation for [c ; known tokens], rotation for [c ; all tokens]); the condition sits at the origin."""
        n_cond = c.size(1) if c is not None else 0
        full = self.rope.get_func(pos, n_cond)
        if ids is None:
            return full, full
        return self.rope.get_func(pos, n_cond, ids.expand(-1, -1, 3)), full

    def _run(self, blocks, x, pe):
        if len(blocks) and use_hip(x) and _plain_attention(blocks[0].attn):
            return _backend.engine().block_stack_forward(list(blocks), x, pe)
        for block in blocks:
            x = block(x, pe)
        return x

    def _final_norm(self, x):
        if not use_hip(x):
            return self.norm(x)
        rows = x.reshape(-1, x.size(-1)).contiguous()
        out = _backend.hip().row_norm(rows, gamma=_f32(self.norm.weight), beta=_f32(self.norm.bias), eps=self.norm.eps)
        return out.view(x.shape)

    def forward(self, x, c=None, prev_ids=None, pos=None) -> torch.Tensor:
        if isinstance(x, (tuple, list)):
            x, prev_ids = x
        if not self.encoder_depth:
            prev_ids = None
        canvas = self.pos_embed(self.patch_embed(x))
        pe_known, pe_full = (None, None) if pos is None else self.prepare_pe(c, prev_ids, pos)
        n_cond = c.size(1) if c is not None else 0
        with_cond = (lambda t: torch.cat([c, t], dim=1)) if n_cond else (lambda t: t)

        # first stage: the condition and the tokens generated so far (in generation order)
        if prev_ids is None:
            x = self._run(self.blocks[: self.encoder_depth], with_cond(canvas), pe_known)
        else:
            index = prev_ids.expand(-1, -1, canvas.size(-1))
            x = self._run(self.blocks[: self.encoder_depth], with_cond(canvas.gather(1, index)), pe_known)
            # second stage input: updated known tokens scattered back onto the mask-token canvas
            tokens = canvas.to(dtype=x.dtype).scatter(1, index, x[:, n_cond:])
            x = torch.cat([x[:, :n_cond], tokens], dim=1) if n_cond else tokens
        x = self._run(self.blocks[self.encoder_depth :], x, pe_full)
        return self._final_norm(x[:, n_cond:] if n_cond else x)
