"""The software-pipelined attention form (variant 5 of nova_debug_set_attn_variant: P V of tile t-1 beside the exponentials of tile t, 64 rows per
wave) of one or two builds against the shipped form, one process, interleaved rounds:  python tools/ab_attn_pipelined.py libA.so [libB.so]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip  # noqa: E402
from microbench import timeit  # noqa: E402

libs = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, argtypes in hip.SIGNATURES.items():
        if hasattr(lib, name):
            getattr(lib, name).argtypes, getattr(lib, name).restype = argtypes, ctypes.c_int
    libs.append((os.path.basename(path), lib))
arms = [(n, lib, v) for n, lib in libs for v in ((3, 5) if lib is libs[0][1] else (5,))]
dt, st = torch.bfloat16, torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
for (S, heads, hd, L) in [(64, 16, 64, 2560), (64, 16, 64, 2049), (64, 16, 64, 1537), (64, 16, 64, 1280)]:
    D = heads * hd
    qkv = torch.randn(S * L, 3 * D, generator=g)
    qkv[:, :D] *= hd ** -0.5 * 1.4426950408889634
    qkv = qkv.to("cuda").to(dt)
    base, res, outs = qkv.data_ptr(), {}, {}
    for _ in range(4):
        for n, lib, var in arms:
            o = outs.setdefault((n, var), torch.empty(S * L, D, dtype=dt, device="cuda"))
            lib.nova_debug_set_attn_variant(var)
            f = lambda: lib.nova_attn_fwd(base, base + 2 * D, base + 4 * D, o.data_ptr(), S, heads, L, L, hd, 3 * D, 3 * D, D, 0.6931471805599453, 1, st)
            res.setdefault((n, var), []).append(timeit(f, iters=6, warm=2))
            lib.nova_debug_set_attn_variant(-1)
    ref = outs[arms[0][0], arms[0][2]].float()
    print(f"L={L}: " + "  ".join(f"{n} variant {v}: {min(t):.3f} ms {4.0 * S * heads * L * L * hd / min(t) / 1e9:5.0f} TF (diff {((outs[n, v].float() - ref).abs().max() / ref.abs().max()).item():.1e})"
                                 for (n, v), t in res.items()), flush=True)
    del qkv, outs
