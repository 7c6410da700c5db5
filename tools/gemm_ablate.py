"""Timing-only ablations of the GEMM256 main loop (variants >= 10 compute garbage)."""
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
for (N, K) in [(1024, 4096), (4096, 1024)]:
    a, w, bias = rnd(M, K), rnd(N, K), torch.randn(N, device="cuda")
    out = torch.empty(M, N, dtype=dt, device="cuda")
    res = {}
    for r in range(3):
        for v in (2, 10, 11, 12):
            hip.call("nova_debug_force_gemm_tile", 2560 + v)
            res.setdefault(v, []).append(timeit(lambda: hip.gemm_bias_act(a, w, bias, 0, out=out), iters=8, warm=2))
    print(f"N={N} K={K}: " + "  ".join(f"v{v}: {min(t):.3f} ms ({2.0 * M * N * K / min(t) / 1e9:5.0f} TF-equiv)" for v, t in res.items()), flush=True)
hip.call("nova_debug_force_gemm_tile", 2562)
hip.call("nova_debug_force_gemm_tile", 0)
