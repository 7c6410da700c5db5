"""What the RoPE epilogue of the fused-QKV GEMM costs: the same GEMM (M = 64 x 2560, N = 3072, K = 1024, bf16) with and
without the rotation, interleaved rounds in one process, plus the plain-bias GEMM of the same shape.
    python tools/qkv_epilogue_cost.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip  # noqa: E402
from microbench import timeit  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
lib = hip.load()
dt = torch.bfloat16
S, L, D, heads = 64, 2560, 1024, 16
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to("cuda").to(dt)
x, w, b = rnd(S * L, D), rnd(3 * D, D), torch.randn(3 * D, device="cuda")
rope = torch.rand(1, L, 32, 2, device="cuda")
rope8 = torch.rand(32, L, 32, 2, device="cuda")  # per-sample tables (first-half encoder blocks: gathered positions)
qkv = torch.empty(S * L, 3 * D, dtype=dt, device="cuda")
st = torch.cuda.current_stream().cuda_stream
runs = {
    "qkv + RoPE (1 table)": lambda: lib.nova_qkv_rope(x.data_ptr(), w.data_ptr(), b.data_ptr(), rope.data_ptr(), qkv.data_ptr(), S, L, D, heads, 1, 1, st),
    "qkv + RoPE (32 tables)": lambda: lib.nova_qkv_rope(x.data_ptr(), w.data_ptr(), b.data_ptr(), rope8.data_ptr(), qkv.data_ptr(), S, L, D, heads, 32, 1, st),
    "qkv, no rotation": lambda: lib.nova_qkv_rope(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, qkv.data_ptr(), S, L, D, heads, 1, 1, st),
    "plain GEMM N=3072": lambda: lib.nova_gemm_bias_act(x.data_ptr(), w.data_ptr(), b.data_ptr(), qkv.data_ptr(), S * L, 3 * D, D, 0, 1, st),
}
res = {k: [] for k in runs}
for _ in range(rounds):
    for k, f in runs.items():
        res[k].append(timeit(f, iters=8, warm=2))
fl = 2.0 * S * L * 3 * D * D
for k, t in res.items():
    t = sorted(t)
    print(f"{k:26s} min {t[0]:.3f} med {t[len(t) // 2]:.3f} ms = {fl / t[len(t) // 2] / 1e9:5.0f} TFLOP/s", flush=True)
