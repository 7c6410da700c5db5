"""Persistent 256x256 GEMM: kernel time vs start stagger (cycles by which the last workgroup's start is delayed)."""
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
hip.call("nova_debug_force_gemm_tile", 2580)
for (N, K, act) in [(1024, 1024, 0), (3072, 1024, 0), (4096, 1024, 1), (1024, 4096, 0)]:
    a, w, bias = rnd(M, K), rnd(N, K), torch.randn(N, device="cuda")
    out = torch.empty(M, N, dtype=dt, device="cuda")
    res = {}
    for rnd_i in range(3):
        for x in (0, 16, 32, 64, 128, 192, 256):
            hip.call("nova_debug_force_gemm_tile", 30000 + x)
            res.setdefault(x, []).append(timeit(lambda: hip.gemm_bias_act(a, w, bias, act, out=out), iters=8, warm=2))
    print(f"N={N} K={K} act={act}: " + "  ".join(f"{x * 256 // 1000}k: {min(t):.3f}ms {2.0 * M * N * K / min(t) / 1e9:5.0f}TF" for x, t in res.items()), flush=True)
    del a, w, out
hip.call("nova_debug_force_gemm_tile", 30000)
hip.call("nova_debug_force_gemm_tile", 0)
