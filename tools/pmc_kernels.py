"""The encoder's seven launch kinds at the metric's shapes, in the product's own composition, for rocprofv3 passes:
one ViT block (nova_vit_blocks_forward: QKV GEMM + RoPE + q-scale, attention, proj GEMM, LN + residual, fc1 GEMM + GELU,
fc2 GEMM, LN + residual) at S = 64 sequences x L = 2560 tokens, D = 1024, 16 heads (config C second-half encoder).

    python3 tools/pmc_kernels.py [width heads L S iters]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nova_pointcloud_amd"))
from nova_pointcloud_amd import engine as E  # noqa: E402
from diffnext.models.vision_transformer import Block  # noqa: E402

D, heads, L, S, iters = (int(v) for v in (sys.argv[1:6] + ["1024", "16", "2560", "64", "3"][len(sys.argv) - 1:]))
torch.manual_seed(0)
blk = Block(D, heads).to("cuda").to(torch.bfloat16).eval()
x = (torch.randn(S, L, D, generator=torch.Generator().manual_seed(1)) * 0.7).to("cuda").to(torch.bfloat16)
rope = torch.rand(1, L, D // heads // 2, 2, device="cuda")


class _PE(object):  # the (cos, sin) table in the layout block_stack_forward reads from a RotaryEmbed3D function object
    weight = torch.stack([torch.stack([rope[..., 0], -rope[..., 1]], -1), torch.stack([rope[..., 1], rope[..., 0]], -1)], -2).unsqueeze(1)


with torch.no_grad():
    for _ in range(iters):
        out = E.block_stack_forward([blk], x, _PE())
torch.cuda.synchronize()
print("pmc_kernels: done", tuple(out.shape), float(out.float().abs().mean()))
