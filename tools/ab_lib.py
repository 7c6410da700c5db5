"""A/B two builds of libnova_hip.so in ONE process (interleaved rounds): python ab_lib.py old.so new.so"""
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
S, L, D, heads = 64, 2560, 1024, 16
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to("cuda").to(dt)
x, w, b = rnd(S * L, D), rnd(3 * D, D), torch.randn(3 * D, device="cuda")
rope = torch.rand(32, L, 32, 2, device="cuda")
qkv = torch.empty(S * L, 3 * D, dtype=dt, device="cuda")
shapes = [("proj N1024 K1024", 1024, 1024, 0), ("fc2 N1024 K4096", 1024, 4096, 0), ("fc1+gelu N4096 K1024", 4096, 1024, 1)]
bufs = {}
for tag, N, K, act in shapes:
    bufs[tag] = (rnd(S * L, K), rnd(N, K), torch.randn(N, device="cuda"), torch.empty(S * L, N, dtype=dt, device="cuda"))
st = torch.cuda.current_stream().cuda_stream
outs = {}
for rnd_i in range(3):
    for name, lib in libs.items():
        f1 = lambda: lib.nova_qkv_rope(x.data_ptr(), w.data_ptr(), b.data_ptr(), rope.data_ptr(), qkv.data_ptr(), S, L, D, heads, 32, 1, st)
        row = [timeit(f1, iters=8, warm=2)]
        for tag, N, K, act in shapes:
            a_, w_, b_, o_ = bufs[tag]
            row.append(timeit(lambda: lib.nova_gemm_bias_act(a_.data_ptr(), w_.data_ptr(), b_.data_ptr(), o_.data_ptr(), S * L, N, K, act, 1, st), iters=8, warm=2))
        outs.setdefault(name, []).append(row + [qkv.float().sum().item(), bufs[shapes[1][0]][3].float().sum().item()])
for name, r in outs.items():
    best = [min(x[i] for x in r) for i in range(4)]
    print(name, "qkv_rope %.3f  " % best[0] + "  ".join("%s %.3f" % (shapes[i][0], best[i + 1]) for i in range(3)), " checksums %.6e %.6e" % (r[0][4], r[0][5]))
