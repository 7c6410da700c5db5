"""torch.autograd bindings of the training kernels of libnova_hip.so.

`attention(q, k, v, attn_mask)` is F.scaled_dot_product_attention (no dropout; no mask, or the block-causal frame mask of multi-frame
training passed as a per-query key limit) for bf16 device tensors with head_dim 64 / 96:
forward on `attn_bf16` with the row log-sum-exp kept, backward on `attn_bwd_dq` / `attn_bwd_dkv` (csrc/attn_bwd.hip;
reference vision_transformer.py:63 under the training forward transformer_3d.py:79-100).

`fused_norm(x, ...)` is the LayerNorm family of the blocks as ONE row kernel forward and ONE backward (csrc/rowops.hip,
csrc/rownorm_bwd.hip): y = LN(x) [* gamma + beta] [* (1 + scale) + shift] [* gate] [+ res] - the post-norm residual of a ViT
block (vision_transformer.py:78-82,91-92), AdaLayerNormZero's modulate (normalization.py:34-36) and DiffusionBlock's gated norm
(diffusion_mlp.py:52-53) - for f32 / bf16 / f16 device tensors. The plain GEMMs and the pointwise activations of the training
step stay on PyTorch's ROCm libraries.
"""
import math
import os

import torch

from . import hip

_ENABLED = os.environ.get("NOVA_TRAIN_ATTN", "1") != "0"  # read once
LOG2E = 1.4426950408889634
stats = {"attention_calls": 0}  # forward calls of the HIP attention in this process (launch tests and logs read it)


_KEY_LIMITS = {}  # id(mask tensor) -> (version, key_limit or None): the O(L^2) structure check runs once per mask


def key_limit_of_mask(attn_mask):
    """int32 [L] `key_limit` with mask[i, j] visible <=> j < key_limit[i] when the [L, L] mask (additive 0 / -inf as the reference builds
    it, embeddings.py:247-260, or boolean) has that form with non-decreasing limits >= 1 - the block-causal frame mask - else None."""
    if attn_mask is None or attn_mask.dim() != 2 or attn_mask.shape[0] != attn_mask.shape[1] or not attn_mask.is_cuda:
        return None
    hit = _KEY_LIMITS.get(id(attn_mask))
    if hit is not None and hit[0] == attn_mask._version and hit[2]() is attn_mask:
        return hit[1]
    import weakref

    L = attn_mask.shape[0]
    allowed = attn_mask if attn_mask.dtype == torch.bool else attn_mask > float("-inf")
    if attn_mask.dtype != torch.bool and not bool(((attn_mask == 0) | ~allowed).all()):
        limit = None  # finite non-zero biases: not a pure visibility mask
    else:
        limit = allowed.sum(-1).to(torch.int32)
        prefix = torch.arange(L, device=attn_mask.device)[None, :] < limit[:, None]
        ok = bool(torch.equal(prefix, allowed)) and bool((limit >= 1).all()) and bool((limit[1:] >= limit[:-1]).all())
        limit = limit.contiguous() if ok else None
    if len(_KEY_LIMITS) > 64:
        _KEY_LIMITS.clear()
    _KEY_LIMITS[id(attn_mask)] = (attn_mask._version, limit, weakref.ref(attn_mask))
    return limit


def attention_supported(q, attn_mask=None):
    """bf16 device tensors [S, heads, L, 64 | 96]; no mask, or a mask of the per-query key-limit form (key_limit_of_mask)."""
    if not (_ENABLED and q.is_cuda and q.dtype == torch.bfloat16 and q.shape[-1] in (64, 96) and q.dim() == 4):
        return False
    return attn_mask is None or (attn_mask.shape[-1] == q.shape[-2] and key_limit_of_mask(attn_mask) is not None)


class NovaAttentionFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, key_limit=None):
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
                 hip.ptr(key_limit), hip.stream_ptr())
        ctx.save_for_backward(qt, kt, vt, o, lse, key_limit)
        ctx.scale = scale
        return o.transpose(1, 2)

    @staticmethod
    def backward(ctx, d_out):
        qt, kt, vt, o, lse, key_limit = ctx.saved_tensors
        S, L, h, d = qt.shape
        do = d_out.transpose(1, 2).to(torch.bfloat16).contiguous()
        delta = torch.empty(S, h, L, dtype=torch.float32, device=do.device)  # filled by the library: sum_c dO * O
        dq, dk, dv = torch.empty_like(qt), torch.empty_like(kt), torch.empty_like(vt)
        hip.call("nova_attn_bwd", qt.data_ptr(), kt.data_ptr(), vt.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
                 delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), S, h, L, d, h * d, h * d, h * d, h * d, ctx.scale,
                 hip.ptr(key_limit), hip.stream_ptr())
        return dq.transpose(1, 2), dk.transpose(1, 2), dv.transpose(1, 2), None


def attention(q, k, v, attn_mask=None):
    """softmax(q k^T / sqrt(d) [+ mask]) v for [S, heads, L, 64 | 96] bf16 device tensors, differentiable. `attn_mask`: None or an
    [L, L] visibility mask of the per-query key-limit form (the block-causal frame mask of multi-frame training)."""
    limit = None
    if attn_mask is not None:
        limit = key_limit_of_mask(attn_mask)
        if limit is None:
            raise ValueError("nova attention: the mask is not of the per-query key-limit form (check attention_supported first)")
    return NovaAttentionFunction.apply(q, k, v, limit)


# ---------------------------------------------------------------------------------------------------------------------
_NORM_ENABLED = os.environ.get("NOVA_TRAIN_NORM", "1") != "0"  # read once
stats["norm_calls"] = 0
_PARTS = 1024  # partial rows of d_gamma / d_beta (= waves of the backward launch)


def _rows2d(t, D):
    """[..., D] view -> [rows, D] view (no copy) or None when the leading dims do not collapse."""
    if t.dim() == 2:
        return t
    try:
        return t.view(-1, D) if t.is_contiguous() else t.flatten(0, -2) if t.flatten(0, -2).data_ptr() == t.data_ptr() else None
    except RuntimeError:
        return None


def _mod_layout(mods, D):
    """scale / shift / gate as the kernels address them: (pointer of the lowest view, row pitch in elements, {name: column offset}).
    They must be [rows, D] views (unit last stride, one common row pitch) of ONE buffer, all inside one row span - what
    `proj(...).chunk(n, dim=-1)` yields (normalization.py:35). None otherwise."""
    views = {k: _rows2d(t, D) for k, t in mods.items()}
    if any(v is None or v.stride(1) != 1 for v in views.values()):
        return None
    first = next(iter(views.values()))
    ld, rows = first.stride(0), first.shape[0]
    vec = 4 if first.dtype == torch.float32 else 8
    if ld % vec or any(v.stride(0) != ld or v.shape[0] != rows or v.untyped_storage().data_ptr() != first.untyped_storage().data_ptr()
                       for v in views.values()):
        return None
    lo = min(v.storage_offset() for v in views.values())
    offs = {k: v.storage_offset() - lo for k, v in views.items()}
    if any(o % vec or o + D > ld for o in offs.values()):
        return None
    ptr = first.untyped_storage().data_ptr() + lo * first.element_size()
    return ptr, ld, offs, rows


def fused_norm_supported(x, gamma=None, scale=None, shift=None, gate=None, res=None):
    """Device rows of width <= 2048 (a multiple of the 16-byte vector), modulation terms given as views of ONE row-major buffer."""
    if not (_NORM_ENABLED and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16, torch.float16)):
        return False
    D = x.shape[-1]
    vec = 4 if x.dtype == torch.float32 else 8
    if D % vec or D > 2048 or (scale is None) != (shift is None):
        return False
    mods = {k: t for k, t in (("scale", scale), ("shift", shift), ("gate", gate)) if t is not None}
    if mods:
        if any(t.dtype != x.dtype or t.shape[-1] != D or t.numel() != x.numel() for t in mods.values()):
            return False
        if _mod_layout(mods, D) is None:
            return False
    return res is None or (res.dtype == x.dtype and res.shape == x.shape)


class NovaFusedNormFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, scale, shift, gate, res, eps):
        hip.load()
        stats["norm_calls"] += 1
        shape, D = x.shape, x.shape[-1]
        for name, t in (("res", res), ("scale", scale), ("shift", shift), ("gate", gate)):
            if t is not None and t.dtype != x.dtype:  # the kernel reads every operand in x's storage type (e.g. autocast: fp16 x, f32 res)
                raise TypeError(f"fused_norm: {name} is {t.dtype}, x is {x.dtype} (see fused_norm_supported)")
        x2 = x.reshape(-1, D).contiguous()
        rows = x2.shape[0]
        mods = {k: t for k, t in (("scale", scale), ("shift", shift), ("gate", gate)) if t is not None}
        mptr, ld, offs = None, 0, {}
        if mods:
            lay = _mod_layout(mods, D)
            if lay is None or lay[3] != rows:
                raise ValueError("fused_norm: scale / shift / gate must be [rows, D] views of one row-major buffer (see fused_norm_supported)")
            mptr, ld, offs, _ = lay
        o = lambda k: offs.get(k, -1)
        res2 = None if res is None else res.reshape(-1, D).contiguous()
        g32 = None if gamma is None else gamma.detach().float().contiguous()
        b32 = None if beta is None else beta.detach().float().contiguous()
        out = torch.empty_like(x2)
        hip.call("nova_row_norm", x2.data_ptr(), out.data_ptr(), hip.ptr(g32), hip.ptr(b32), mptr, ld, o("scale"), o("shift"), o("gate"),
                 hip.ptr(res2), None, rows, D, float(eps), hip.dtype_code(x.dtype), hip.stream_ptr())
        ctx.save_for_backward(x2, g32, b32, *mods.values())  # the views keep the modulation buffer alive (and versioned)
        ctx.meta = (shape, tuple(mods), float(eps), None if gamma is None else gamma.dtype, res is not None)
        return out.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, g32, b32, *mod_t = ctx.saved_tensors
        shape, names, eps, gdtype, has_res = ctx.meta
        D, rows = shape[-1], x2.shape[0]
        mods = dict(zip(names, mod_t))
        mptr, ld, offs, dmod = None, 0, {}, None
        if mods:
            mptr, ld, offs, _ = _mod_layout(mods, D)
            dmod = torch.empty(rows, ld, dtype=x2.dtype, device=x2.device)
        o = lambda k: offs.get(k, -1)
        dy2 = dy.reshape(-1, D).to(x2.dtype).contiguous()
        dx = torch.empty_like(x2)
        parts = min(_PARTS, max(4, ((rows + 3) // 4) * 4))
        dgp = dbp = None
        if g32 is not None:
            dgp = torch.empty(parts, D, dtype=torch.float32, device=x2.device)
            dbp = torch.empty(parts, D, dtype=torch.float32, device=x2.device)
        hip.call("nova_row_norm_bwd", x2.data_ptr(), dy2.data_ptr(), hip.ptr(g32), hip.ptr(b32), mptr, ld, o("scale"), o("shift"), o("gate"),
                 dx.data_ptr(), hip.ptr(dmod), hip.ptr(dgp), hip.ptr(dbp), parts, rows, D, eps, hip.dtype_code(x2.dtype), hip.stream_ptr())
        grad = lambda k: dmod[:, offs[k]:offs[k] + D].reshape(mods[k].shape) if k in mods else None
        return (dx.view(shape), None if g32 is None else dgp.sum(0).to(gdtype), None if g32 is None else dbp.sum(0).to(gdtype),
                grad("scale"), grad("shift"), grad("gate"), dy if has_res else None, None)


def fused_norm(x, gamma=None, beta=None, scale=None, shift=None, gate=None, res=None, eps=1e-5):
    """y = LN(x; eps) [* gamma + beta] [* (1 + scale) + shift] [* gate] [+ res], differentiable; one HIP row kernel each way."""
    return NovaFusedNormFunction.apply(x, gamma, beta, scale, shift, gate, res, eps)


# ---- pointwise activations of the MLPs (csrc/rownorm_bwd.hip: act_kernel) ----------------------------------------------------
_ACT_ENABLED = os.environ.get("NOVA_TRAIN_ACT", "1") != "0"  # read once
ACT_GELU, ACT_SILU = 1, 2  # nova_act of include/nova_hip.h
stats["act_calls"] = 0


def activation_supported(x):
    """GPU tensor of a storage type the library knows, whole 16-byte chunks (every width of the model is a multiple of 8)."""
    if not (_ACT_ENABLED and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16, torch.float16)):
        return False
    return x.numel() > 0 and x.numel() % (4 if x.dtype == torch.float32 else 8) == 0


class NovaActivationFunction(torch.autograd.Function):
    """y = gelu(x) (exact erf form, nn.GELU()) or silu(x); the backward recomputes the slope from x (nothing else is saved)."""

    @staticmethod
    def forward(ctx, x, kind):
        hip.load()
        stats["act_calls"] += 1
        xc = x.contiguous()
        y = torch.empty_like(xc)
        hip.call("nova_act_fwd", xc.data_ptr(), y.data_ptr(), xc.numel(), int(kind), hip.dtype_code(xc.dtype), hip.stream_ptr())
        ctx.save_for_backward(xc)
        ctx.kind = int(kind)
        return y

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        dyc = dy.to(xc.dtype).contiguous()
        dx = torch.empty_like(xc)
        hip.call("nova_act_bwd", xc.data_ptr(), dyc.data_ptr(), dx.data_ptr(), xc.numel(), ctx.kind, hip.dtype_code(xc.dtype), hip.stream_ptr())
        return dx, None


def activation(x, kind):
    """gelu / silu of `x`, differentiable; one HIP pointwise kernel each way (ACT_GELU / ACT_SILU)."""
    return NovaActivationFunction.apply(x, kind)
