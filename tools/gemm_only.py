"""One large GEMM launch shape in isolation (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip  # noqa: E402

M, N, K = 64 * 2560, 1024, 4096
g = torch.Generator().manual_seed(0)
a = (torch.randn(M, K, generator=g) * 0.5).to("cuda").to(torch.bfloat16)
w = (torch.randn(N, K, generator=g) * 0.5).to("cuda").to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(4):
    hip.gemm_bias_act(a, w, None, 0, out=out)
torch.cuda.synchronize()
