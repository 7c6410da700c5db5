"""Training kernels of libnova_hip.so against torch autograd (GPU box only).

Attention backward (csrc/attn_bwd.hip): forward output and dq / dk / dv of `nova_pointcloud_amd.autograd.attention` against
F.scaled_dot_product_attention differentiated in float32 on the same bf16-valued inputs. Tolerance: P and dS pass
through bf16 on their way into the MFMAs (2^-8 relative per element), gradients are sums of such terms - 2e-2 of the
tensor's largest magnitude.
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nova_pointcloud_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)  # the drop-in `diffnext` package


def _ref(q, k, v, d_out):
    qf, kf, vf = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    out = torch.nn.functional.scaled_dot_product_attention(qf, kf, vf)
    out.backward(d_out.float())
    return out.detach(), qf.grad, kf.grad, vf.grad


def _rel(got, ref):
    return ((got.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()


@pytest.mark.parametrize("S,h,L", [(1, 1, 64), (2, 3, 128), (1, 2, 200), (2, 1, 333), (1, 4, 1031), (3, 2, 31)])
@pytest.mark.parametrize("spread", [1.0, 4.0])
def test_attention_forward_backward_match_autograd(hip, S, h, L, spread):
    from nova_pointcloud_amd import autograd as A

    g = torch.Generator().manual_seed(S * 1000 + h * 100 + L)
    mk = lambda s: (torch.randn(S, h, L, 64, generator=g) * s).bfloat16().cuda()
    q, k, v, d_out = mk(spread), mk(1.0), mk(1.0), mk(1.0)  # spread > 1: peaked rows (large score range)
    q, k, v = (t.requires_grad_(True) for t in (q, k, v))
    assert A.attention_supported(q)
    out = A.attention(q, k, v)
    out.backward(d_out)
    o_ref, dq_ref, dk_ref, dv_ref = _ref(q, k, v, d_out)
    assert out.shape == (S, h, L, 64) and q.grad.shape == q.shape
    assert _rel(out, o_ref) < 1.6e-2
    for name, got, ref in (("dq", q.grad, dq_ref), ("dk", k.grad, dk_ref), ("dv", v.grad, dv_ref)):
        assert torch.isfinite(got.float()).all(), name
        assert _rel(got, ref) < 2e-2, (name, _rel(got, ref))


def test_attention_module_trains_through_hip_kernels(hip):
    """`Attention.forward` with autograd on (the training forward) takes the HIP attention for bf16 / head_dim 64 and its
    parameter gradients match the PyTorch definition of the same module."""
    from diffnext.models.vision_transformer import Attention
    from nova_pointcloud_amd import autograd as A

    torch.manual_seed(5)
    attn = Attention(256, 4).cuda().bfloat16()
    x = (torch.randn(2, 150, 256) * 0.7).bfloat16().cuda()
    calls = []
    orig = A.NovaAttentionFunction.apply
    try:
        A.NovaAttentionFunction.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
        xa = x.clone().requires_grad_(True)
        attn(xa).float().square().mean().backward()
    finally:
        A.NovaAttentionFunction.apply = orig
    assert calls, "training forward did not reach the HIP attention"
    got = {n: p.grad.float().clone() for n, p in attn.named_parameters()}
    gx = xa.grad.float().clone()
    attn.zero_grad()
    ref_mod = Attention(256, 4).cuda().float()
    ref_mod.load_state_dict({k_: v_.float() for k_, v_ in attn.state_dict().items()})
    xb = x.float().requires_grad_(True)
    ref_mod(xb).square().mean().backward()
    assert _rel(gx, xb.grad) < 3e-2
    for n, p in ref_mod.named_parameters():
        assert _rel(got[n], p.grad) < 3e-2, n
