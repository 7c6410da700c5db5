"""A/B two builds of libnova_hip.so on the attention kernel in ONE process: python ab_attn.py old.so new.so"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip  # noqa: E402
from microbench import timeit  # noqa: E402

libs = {}
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, argtypes in hip.SIGNATURES.items():
        if hasattr(lib, name):
            getattr(lib, name).argtypes, getattr(lib, name).restype = argtypes, ctypes.c_int
    libs[os.path.basename(path)] = lib
dt = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
for (S, heads, hd, L) in [(64, 16, 64, 2560), (64, 16, 64, 768), (64, 16, 64, 1280), (64, 16, 64, 1537), (64, 16, 64, 2049), (64, 16, 64, 2510), (64, 16, 96, 2560), (64, 16, 96, 1537)]:
    D = heads * hd
    qkv = torch.randn(S * L, 3 * D, generator=g)
    qkv[:, :D] *= hd ** -0.5 * 1.4426950408889634  # what the fused QKV epilogue delivers
    qkv = qkv.to("cuda").to(dt)
    outs, res = {}, {}
    for rnd_i in range(3):
        for name, lib in libs.items():
            o = outs.setdefault(name, torch.empty(S * L, D, dtype=dt, device="cuda"))
            base = qkv.data_ptr()
            f = lambda: lib.nova_attn_fwd(base, base + 2 * D, base + 4 * D, o.data_ptr(), S, heads, L, L, hd, 3 * D, 3 * D, D,
                                          0.6931471805599453, 1, st)  # scale * log2(e) == 1.0f: q counts as pre-scaled
            res.setdefault(name, []).append(timeit(f, iters=6, warm=2))
    names = list(libs)
    same = ((outs[names[0]].float() - outs[names[-1]].float()).abs().max() / outs[names[0]].float().abs().max()).item()
    print(f"hd={hd} L={L}: " + "  ".join(f"{n}: {min(t):.3f} ms {4.0 * S * heads * L * L * hd / min(t) / 1e9:5.0f} TF" for n, t in res.items()) + f"  max rel diff first-vs-last {same:.2e}", flush=True)
    del qkv, outs
