"""One attention launch shape in isolation (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip  # noqa: E402

S, heads, L = 64, 16, 2560
D = heads * 64
g = torch.Generator().manual_seed(0)
qkv = torch.randn(S * L, 3 * D, generator=g).to("cuda").to(torch.bfloat16)
o = torch.empty(S * L, D, dtype=torch.bfloat16, device="cuda")
for _ in range(4):
    hip.attn_fwd_packed(qkv, S, L, heads, out=o)
torch.cuda.synchronize()
