"""128-tile vs persistent 256-tile GEMM at decoder-sized M (rows = sequences x tokens predicted in an AR step)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip  # noqa: E402
from microbench import timeit  # noqa: E402

dt = torch.bfloat16
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to("cuda").to(dt)
import sys as _sys
cases = [(20480, 1024, 0, (512, 1216, 2048, 3072, 4096, 6272)), (1024, 1024, 2, (512, 2048, 4096, 6272)),
         (1024, 1024, 0, (4096, 6272, 9600, 12800, 20480, 40960))]
if len(_sys.argv) > 1 and _sys.argv[1] == "w768":  # config B (d48w768, 16 sequences)
    cases = [(15360, 768, 0, (320, 800, 1600)), (768, 768, 0, (4608, 7296, 12288, 20480)), (3072, 768, 1, (4608, 7296, 20480)),
             (768, 3072, 0, (4608, 7296, 20480)), (2304, 768, 0, (4608, 7296, 20480))]
for (N, K, act, Ms) in cases:
    w, bias = rnd(N, K), torch.randn(N, device="cuda")
    for M in Ms:
        a = rnd(M, K)
        out = torch.empty(M, N, dtype=dt, device="cuda")
        res = {}
        for r in range(3):
            for tile in (128, 2580):
                hip.call("nova_debug_force_gemm_tile", tile)
                res.setdefault(tile, []).append(timeit(lambda: hip.gemm_bias_act(a, w, bias, act, out=out), iters=10, warm=2))
        print(f"N={N} K={K} act={act} M={M}: " + "  ".join(f"{t}: {min(v) * 1e3:7.1f} us {2.0 * M * N * K / min(v) / 1e9:5.0f} TF" for t, v in res.items()), flush=True)
hip.call("nova_debug_force_gemm_tile", 0)
