"""Per-tile fixed cost of the 256x256 GEMM structures: time vs K at fixed M, N (linear fit a + b * K-tiles)."""
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
for N in (1024, 4096):
    tiles_per_cu = (M // 256) * (N // 256) / 256
    for v in (2, 20):
        pts = []
        for K in (256, 512, 1024, 2048, 4096):
            a, w, bias = rnd(M, K), rnd(N, K), torch.randn(N, device="cuda")
            out = torch.empty(M, N, dtype=dt, device="cuda")
            hip.call("nova_debug_force_gemm_tile", 2560 + v)
            ms = min(timeit(lambda: hip.gemm_bias_act(a, w, bias, 0, out=out), iters=8, warm=2) for _ in range(3))
            pts.append((K // 64, ms * 1e3 / tiles_per_cu))  # us per tile
            del a, w, out
        (k0, t0), (k1, t1) = pts[0], pts[-1]
        b = (t1 - t0) / (k1 - k0)
        print(f"N={N} v{v}: " + "  ".join(f"K={k * 64}: {t:.1f}us" for k, t in pts) + f"   slope {b:.3f} us/K-tile, intercept {t0 - b * k0:.1f} us", flush=True)
hip.call("nova_debug_force_gemm_tile", 0)
