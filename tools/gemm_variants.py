"""A/B of the GEMM256 schedule variants in one process (interleaved rounds), bitwise-checked against the 128 tile."""
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
shapes = [(1024, 1024, 0), (1024, 4096, 0), (4096, 1024, 1), (3072, 1024, 0)]  # proj, fc2, fc1 (GELU), qkv-sized
for (N, K, act) in shapes:
    a, w, bias = rnd(M, K), rnd(N, K), torch.randn(N, device="cuda")
    hip.call("nova_debug_force_gemm_tile", 128)
    ref = hip.gemm_bias_act(a, w, bias, act)
    out = torch.empty_like(ref)
    res = {}
    for rnd_i in range(3):
        for v in (2, 20):
            hip.call("nova_debug_force_gemm_tile", 2560 + v)
            ms = timeit(lambda: hip.gemm_bias_act(a, w, bias, act, out=out), iters=8, warm=2)
            assert torch.equal(out, ref), f"variant {v} differs from the 128 tile"
            res.setdefault(v, []).append(ms)
    line = "  ".join(f"v{v}: {min(t):.3f} ms {2.0 * M * N * K / min(t) / 1e9:6.0f} TF" for v, t in res.items())
    print(f"N={N} K={K} act={act}: {line}", flush=True)
    del a, w, ref, out
# fused QKV + RoPE (+ q scale) at the encoder shape
S, L, D, heads = 64, 2560, 1024, 16
x, w, b = rnd(S * L, D), rnd(3 * D, D), torch.randn(3 * D, device="cuda")
rope = torch.rand(32, L, 32, 2, device="cuda")
hip.call("nova_debug_force_gemm_tile", 128)
ref = hip.qkv_rope(x, w, b, rope, S, L, heads)
out = torch.empty_like(ref)
res = {}
for rnd_i in range(3):
    for v in (2, 20):
        hip.call("nova_debug_force_gemm_tile", 2560 + v)
        ms = timeit(lambda: hip.qkv_rope(x, w, b, rope, S, L, heads, out=out), iters=8, warm=2)
        assert torch.equal(out, ref), f"variant {v} differs from the 128 tile"
        res.setdefault(v, []).append(ms)
print("qkv+rope: " + "  ".join(f"v{v}: {min(t):.3f} ms {2.0 * S * L * 3 * D * D / min(t) / 1e9:6.0f} TF" for v, t in res.items()), flush=True)
hip.call("nova_debug_force_gemm_tile", 0)
