"""Small-M GEMM (skinny.hip) against the 128-tile kernel at the diffusion MLP's per-step shapes, and the fused
modulate -> fc1 -> SiLU launch against row_norm + GEMM. Back-to-back launches of one shape on one stream, HIP events:
the per-launch figure includes the launch gap, which is what the denoising loop pays.

    python3 tools/skinny_bench.py            # D = 768 and 1024, M = 16 .. 4096
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip  # noqa: E402
from microbench import timeit  # noqa: E402

dt = torch.bfloat16
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to("cuda").to(dt)


def force(tile):
    hip.call("nova_debug_force_gemm_tile", tile)


for D in (768, 1024):
    w, bias = rnd(D, D), torch.randn(D, device="cuda")
    for M in (16, 64, 128, 256, 400, 512, 800, 1024, 1600, 2048, 3264, 4096):
        a, mod = rnd(M, D), rnd(M, 20 * D)
        out, h = torch.empty(M, D, dtype=dt, device="cuda"), torch.empty(M, D, dtype=dt, device="cuda")
        t = {}
        for tile in (16, 128):
            force(tile)
            t[tile, "gemm"] = min(timeit(lambda: hip.gemm_bias_act(a, w, bias, 2, out=out), iters=50, warm=5) for _ in range(3))
            t[tile, "adaln"] = min(timeit(lambda: hip.adaln_fc1(a, mod, D, 2 * D, w, bias, out=out), iters=50, warm=5) for _ in range(3))
        force(0)
        auto = min(timeit(lambda: hip.adaln_fc1(a, mod, D, 2 * D, w, bias, out=out), iters=50, warm=5) for _ in range(3))
        print(f"D={D} M={M:5d}: gemm small-M {t[16, 'gemm'] * 1e3:6.1f} us  128-tile {t[128, 'gemm'] * 1e3:6.1f} us | "
              f"modulate+fc1 fused {t[16, 'adaln'] * 1e3:6.1f} us  two launches {t[128, 'adaln'] * 1e3:6.1f} us  auto {auto * 1e3:6.1f} us",
              flush=True)
