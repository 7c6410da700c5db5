"""Per-shape timing of the hot kernels on one MI355X (HIP events on the current stream)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nova_pointcloud_amd import hip  # noqa: E402


def use_experiments_lib():
    """Point the binding at libnova_hip_exp.so (`make -C nova_pointcloud_amd/csrc exp`): the -DNOVA_EXPERIMENTS build
    with the A/B schedule variants and the timing-only ablation kernels. Call before the first hip.load()."""
    path = os.path.join(ROOT, "nova_pointcloud_amd", "libnova_hip_exp.so")
    if not os.path.exists(path):
        raise SystemExit(f"{path} not found: run `make -C nova_pointcloud_amd/csrc exp` first")
    hip._LIB_PATH = path


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dt = torch.bfloat16
    M = 64 * 2560
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to("cuda").to(dt)
    print("GEMM (M=%d, bf16):" % M)
    for tile in (256, 128):
        hip.call("nova_debug_force_gemm_tile", tile)
        for (N, K, act) in [(1024, 1024, 0), (1024, 2048, 0), (1024, 4096, 0), (4096, 1024, 0), (4096, 1024, 1), (3072, 1024, 0)]:
            a, w, bias = rnd(M, K), rnd(N, K), torch.randn(N, device="cuda")
            out = torch.empty(M, N, dtype=dt, device="cuda")
            ms = timeit(lambda: hip.gemm_bias_act(a, w, bias, act, out=out))
            print(f"  tile{tile} N={N:5d} K={K:5d} act={act}: {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s")
            del a, w, out
    hip.call("nova_debug_force_gemm_tile", 0)
    S, L, D, heads = 64, 2560, 1024, 16
    x, w, b = rnd(S * L, D), rnd(3 * D, D), torch.randn(3 * D, device="cuda")
    rope = torch.rand(32, L, 32, 2, device="cuda")
    qkv = torch.empty(S * L, 3 * D, dtype=dt, device="cuda")
    ms = timeit(lambda: hip.qkv_rope(x, w, b, rope, S, L, heads, out=qkv))
    print(f"QKV+RoPE: {ms:7.3f} ms  {2.0 * S * L * 3 * D * D / ms / 1e9:7.1f} TFLOP/s")
    o = torch.empty(S * L, D, dtype=dt, device="cuda")
    for Lx in (2560, 1536, 512):
        q = qkv[: S * Lx]
        ms = timeit(lambda: hip.attn_fwd_packed(q, S, Lx, heads, out=o[: S * Lx]))
        print(f"attention L={Lx}: {ms:7.3f} ms  {4.0 * S * heads * Lx * Lx * 64 / ms / 1e9:7.1f} TFLOP/s")
    res = rnd(S * L, D)
    gam, bet = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    ms = timeit(lambda: hip.row_norm(o, out=res, gamma=gam, beta=bet, res=res))
    print(f"row_norm+res: {ms:7.3f} ms  {3.0 * S * L * D * 2 / ms / 1e6:7.1f} GB/s")


if __name__ == "__main__":
    main()
