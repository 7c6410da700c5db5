"""Attention forward + backward kernels (csrc/attn.hip, attn_bwd.hip) at the encoder's shapes against PyTorch's SDPA
forward + backward on the same tensors: milliseconds and TFLOP/s (forward 4 S h L^2 d, backward as executed here
14 S h L^2 d - seven MFMA products - and 10 S h L^2 d algorithmic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import autograd as A, hip  # noqa: E402

if os.environ.get("NOVA_HIP_LIB"):  # A/B another build of the library
    hip._LIB_PATH = os.environ["NOVA_HIP_LIB"]
from microbench import timeit  # noqa: E402

for (S, h, L) in ((16, 16, 2560), (16, 12, 1280), (4, 16, 2560)):
    g = torch.Generator().manual_seed(0)
    q, k, v, do = ((torch.randn(S, h, L, 64, generator=g)).bfloat16().cuda() for _ in range(4))
    flop = 4.0 * S * h * L * L * 64

    def run(fn):
        qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
        out = fn(qq, kk, vv)
        torch.cuda.synchronize()
        fwd = timeit(lambda: fn(qq, kk, vv), iters=5, warm=2)
        both = timeit(lambda: fn(qq, kk, vv).backward(do), iters=5, warm=2)
        return fwd, both - fwd

    f_hip, b_hip = run(A.attention)
    f_pt, b_pt = run(torch.nn.functional.scaled_dot_product_attention)
    print(f"S={S} h={h} L={L}: forward HIP {f_hip:.3f} ms ({flop / f_hip / 1e9:.0f} TF incl. the layout copies)  torch {f_pt:.3f} ms | "
          f"backward HIP {b_hip:.3f} ms ({2.5 * flop / b_hip / 1e9:.0f} TF algorithmic)  torch {b_pt:.3f} ms ({2.5 * flop / b_pt / 1e9:.0f} TF)", flush=True)
