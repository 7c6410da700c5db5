"""A/B of the attention softmax variants (interleaved rounds in one process) + numerics vs torch SDPA."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip  # noqa: E402
from microbench import timeit  # noqa: E402

dt = torch.bfloat16
S, heads = 64, 16
D = heads * 64
g = torch.Generator().manual_seed(0)
for L in (2560, 1537):
    qkv = (torch.randn(S * L, 3 * D, generator=g)).to("cuda").to(dt)
    o = torch.empty(S * L, D, dtype=dt, device="cuda")
    q, k, v = qkv[: 2 * L].float().view(2, L, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(2 * L, D)
    res = {}
    for rnd_i in range(3):
        for var in (0, 1):
            hip.call("nova_debug_force_gemm_tile", 9000 + var)
            ms = timeit(lambda: hip.attn_fwd_packed(qkv, S, L, heads, out=o), iters=6, warm=2)
            err = ((o[: 2 * L].float() - ref).abs().max() / ref.abs().max()).item()
            res.setdefault(var, []).append((ms, err))
    print(f"L={L}: " + "  ".join(f"v{v}: {min(t for t, _ in r):.3f} ms {4.0 * S * heads * L * L * 64 / min(t for t, _ in r) / 1e9:5.0f} TF err {r[0][1]:.2e}"
                                 for v, r in res.items()), flush=True)
hip.call("nova_debug_force_gemm_tile", 9000)
