"""torch.autograd bindings of the training kernels of libnova_hip.so.

`attention(q, k, v)` is F.scaled_dot_product_attention (no mask, no dropout) for bf16 device tensors with head_dim 64 / 96:
forward on `attn_bf16` with the row log-sum-exp kept, backward on `attn_bwd_dq` / `attn_bwd_dkv` (csrc/attn_bwd.hip;
reference vision_transformer.py:63 under the training forward transformer_3d.py:79-100). Everything else of the
training step stays on PyTorch's ROCm libraries (plain GEMMs, LayerNorm, GELU).
"""
import math
import os

import torch

from . import hip

_ENABLED = os.environ.get("NOVA_TRAIN_ATTN", "1") != "0"  # read once
LOG2E = 1.4426950408889634
stats = {"attention_calls": 0}  # forward calls of the HIP attention in this process (launch tests and logs read it)


def attention_supported(q, attn_mask=None):
    return (_ENABLED and q.is_cuda and q.dtype == torch.bfloat16 and q.shape[-1] in (64, 96) and attn_mask is None and q.dim() == 4)


class NovaAttentionFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v):
        S, h, L, d = q.shape
        if k.shape != q.shape or v.shape != q.shape:
            raise ValueError("nova attention (training) is self-attention: q, k, v must share one shape [S, heads, L, head_dim]")
        hip.load()
        stats["attention_calls"] += 1
        scale = 1.0 / math.sqrt(d)
        # token-major copies [S, L, h, d]; q pre-scaled into the exp2 domain exactly as the generation path's QKV epilogue does
        qt = (q.transpose(1, 2).float() * (scale * LOG2E)).to(torch.bfloat16).contiguous()
        kt, vt = k.transpose(1, 2).contiguous(), v.transpose(1, 2).contiguous()
        o = torch.empty(S, L, h, d, dtype=torch.bfloat16, device=q.device)
        lse = torch.empty(S, h, L, dtype=torch.float32, device=q.device)
        hip.call("nova_attn_fwd_lse", qt.data_ptr(), kt.data_ptr(), vt.data_ptr(), o.data_ptr(), lse.data_ptr(), S, h, L, d, h * d, h * d,
                 hip.stream_ptr())
        ctx.save_for_backward(qt, kt, vt, o, lse)
        ctx.scale = scale
        return o.transpose(1, 2)

    @staticmethod
    def backward(ctx, d_out):
        qt, kt, vt, o, lse = ctx.saved_tensors
        S, L, h, d = qt.shape
        do = d_out.transpose(1, 2).to(torch.bfloat16).contiguous()
        delta = torch.empty(S, h, L, dtype=torch.float32, device=do.device)  # filled by the library: sum_c dO * O
        dq, dk, dv = torch.empty_like(qt), torch.empty_like(kt), torch.empty_like(vt)
        hip.call("nova_attn_bwd", qt.data_ptr(), kt.data_ptr(), vt.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
                 delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), S, h, L, d, h * d, h * d, h * d, h * d, ctx.scale,
                 hip.stream_ptr())
        return dq.transpose(1, 2), dk.transpose(1, 2), dv.transpose(1, 2)


def attention(q, k, v):
    """softmax(q k^T / sqrt(d)) v for [S, heads, L, 64 | 96] bf16 device tensors, differentiable."""
    return NovaAttentionFunction.apply(q, k, v)
