"""A/B of the GEMM256 tile-group shape (row panels per group) in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip  # noqa: E402
from microbench import timeit, use_experiments_lib  # noqa: E402

use_experiments_lib()

dt = torch.bfloat16
M = 64 * 2560
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to("cuda").to(dt)
for (N, K, act) in [(1024, 1024, 0), (1024, 4096, 0), (4096, 1024, 1), (3072, 1024, 0)]:
    a, w, bias = rnd(M, K), rnd(N, K), torch.randn(N, device="cuda")
    out = torch.empty(M, N, dtype=dt, device="cuda")
    res = {}
    for r in range(3):
        for gm in (1, 2, 4, 8, 16, 32):
            hip.call("nova_debug_force_gemm_tile", 7000 + gm)
            res.setdefault(gm, []).append(timeit(lambda: hip.gemm_bias_act(a, w, bias, act, out=out), iters=8, warm=2))
    print(f"N={N} K={K} act={act}: " + "  ".join(f"gm{k}: {min(t):.3f}" for k, t in res.items()), flush=True)
hip.call("nova_debug_force_gemm_tile", 7008)
