"""Access to the MI355X backend (nova_pointcloud_amd.hip / .engine) from the `diffnext` package.

`diffnext` can be imported either as `nova_pointcloud_amd.diffnext` or, when
`<repo>/nova_pointcloud_amd` is put on PYTHONPATH to shadow the reference, as top-level
`diffnext`. In both cases the backend is the single module pair `nova_pointcloud_amd.hip` /
`nova_pointcloud_amd.engine`; this file makes sure its parent directory is importable.
"""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _module(name):
    try:
        return importlib.import_module("nova_pointcloud_amd." + name)
    except ModuleNotFoundError as e:
        if e.name not in ("nova_pointcloud_amd", "nova_pointcloud_amd." + name):
            raise
        sys.path.append(_ROOT)
        return importlib.import_module("nova_pointcloud_amd." + name)


def hip():
    """ctypes binding of libnova_hip.so; loading fails loudly when the library is not built."""
    return _module("hip")


def engine():
    return _module("engine")


def autograd():
    """torch.autograd bindings of the training kernels (nova_pointcloud_amd/autograd.py)."""
    return _module("autograd")


def train_attention_supported(q, attn_mask):
    """bf16 device tensors [S, heads, L, 64 | 96] without a mask or with the block-causal frame mask: what the HIP attention
    forward + backward are built for."""
    return q.is_cuda and autograd().attention_supported(q, attn_mask)


def train_norm_supported(x, **terms):
    """Training on the GPU (autograd on): the LayerNorm family as one HIP row kernel each way (autograd.fused_norm)."""
    import torch

    return x.is_cuda and torch.is_grad_enabled() and autograd().fused_norm_supported(x, **terms)


def train_activation(x, kind, fallback):
    """Training on the GPU (autograd on): gelu (kind 1, the exact erf form) / silu (kind 2) as one HIP pointwise kernel each way
    (autograd.activation, csrc/rownorm_bwd.hip); everywhere else `fallback(x)`, the torch op of the reference module."""
    import torch

    if x.is_cuda and torch.is_grad_enabled() and autograd().activation_supported(x):
        return autograd().activation(x, kind)
    return fallback(x)
