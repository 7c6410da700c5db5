"""MX-fp8 GEMM (v_mfma_scale_f32_16x16x128_f8f6f4) against the bf16 persistent GEMM at encoder shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip  # noqa: E402
from microbench import timeit  # noqa: E402

M = 64 * 2560
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to("cuda").to(torch.bfloat16)
for (N, K, act) in [(1024, 1024, 0), (1024, 4096, 0), (4096, 1024, 1), (3072, 1024, 0), (1536, 1536, 0), (6144, 1536, 1), (1536, 6144, 0)]:
    a, w, bias = rnd(M, K), rnd(N, K), torch.randn(N, device="cuda")
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    a8, sa = hip.quantize_rows_fp8(a)
    w8, sw = hip.quantize_rows_fp8(w)
    tb = min(timeit(lambda: hip.gemm_bias_act(a, w, bias, act, out=out), iters=8, warm=2) for _ in range(3))
    t8 = min(timeit(lambda: hip.gemm_fp8_bias_act(a8, sa, w8, sw, bias, act, out=out), iters=8, warm=2) for _ in range(3))
    hip.call("nova_debug_force_gemm_tile", 258)  # the persistent kernels in their per-tile-prologue form (rounds 1-3)
    tbp = min(timeit(lambda: hip.gemm_bias_act(a, w, bias, act, out=out), iters=8, warm=2) for _ in range(3))
    t8p = min(timeit(lambda: hip.gemm_fp8_bias_act(a8, sa, w8, sw, bias, act, out=out), iters=8, warm=2) for _ in range(3))
    hip.call("nova_debug_force_gemm_tile", 0)
    tq = min(timeit(lambda: hip.quantize_rows_fp8(a), iters=8, warm=2) for _ in range(3))
    fl = 2.0 * M * N * K
    print(f"N={N} K={K} act={act}: bf16 {tb:.3f} ms {fl / tb / 1e9:5.0f} TF | fp8 {t8:.3f} ms {fl / t8 / 1e9:5.0f} TF ({tb / t8:.2f}x) | "
          f"prologue form: bf16 {tbp:.3f} fp8 {t8p:.3f} ms | quantize A {tq:.3f} ms", flush=True)
    del a, w, out, a8, w8
