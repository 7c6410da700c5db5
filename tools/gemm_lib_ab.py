"""Two builds of the library against each other on the encoder's four GEMM launches (+ the LayerNorm pass that reads the out-projection's
result), interleaved in ONE process on one device (cdna_hip_programming.md rule 24): for compile-time experiments.

    python tools/gemm_lib_ab.py nova_pointcloud_amd/libnova_hip.so build_exp/libnova_<variant>.so
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip  # noqa: E402


def bind(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, argtypes in hip.SIGNATURES.items():
        if hasattr(lib, name):
            getattr(lib, name).argtypes, getattr(lib, name).restype = argtypes, ctypes.c_int
    return lib


libs = [bind(p) for p in sys.argv[1:3]]
dt = torch.bfloat16
S, L, D, heads = 64, 2560, 1024, 16
g = torch.Generator().manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to("cuda").to(dt)
st = torch.cuda.current_stream().cuda_stream
ROUNDS, ITERS = 7, 10


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn()
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS


def ab(tag, make, flop=None):
    res = [[], []]
    for _ in range(ROUNDS):
        for i, lib in enumerate(libs):
            res[i].append(timed(make(lib)))
    a, b = np.median(res[0]), np.median(res[1])
    rate = f"  {flop / a * 1e-9:7.1f} -> {flop / b * 1e-9:7.1f} TFLOP/s" if flop else ""
    print(f"{tag:40s} A {a:7.4f} ms (min {min(res[0]):7.4f})   B {b:7.4f} ms (min {min(res[1]):7.4f})   B/A {b / a:5.3f}{rate}", flush=True)


x, w, b = rnd(S * L, D), rnd(3 * D, D, scale=D ** -0.5), torch.randn(3 * D, device="cuda")
rope = torch.rand(2, L, 32, 2, device="cuda")
qkv = torch.empty(S * L, 3 * D, dtype=dt, device="cuda")
ab("QKV + RoPE 163840 x 3072 x 1024", lambda lib: lambda: lib.nova_qkv_rope(x.data_ptr(), w.data_ptr(), b.data_ptr(), rope.data_ptr(), qkv.data_ptr(), S, L, D, heads, 2, 1, st), 2.0 * S * L * 3 * D * D)
for tag, N, K, act in (("out-projection 163840 x 1024 x 1024", 1024, 1024, 0), ("fc1 + GELU 163840 x 4096 x 1024", 4096, 1024, 1), ("fc2 163840 x 1024 x 4096", 1024, 4096, 0)):
    a_, w_, b_, o_ = rnd(S * L, K), rnd(N, K, scale=K ** -0.5), torch.randn(N, device="cuda"), torch.empty(S * L, N, dtype=dt, device="cuda")
    ab(tag, lambda lib: lambda: lib.nova_gemm_bias_act(a_.data_ptr(), w_.data_ptr(), b_.data_ptr(), o_.data_ptr(), S * L, N, K, act, 1, st), 2.0 * S * L * N * K)
    if N == 1024 and K == 1024:  # the pair the encoder runs: out-projection, then the LayerNorm + residual pass over its result
        res_, out_ = rnd(S * L, N), torch.empty(S * L, N, dtype=dt, device="cuda")
        gam, bet = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")

        def pair(lib):
            def run():
                lib.nova_gemm_bias_act(a_.data_ptr(), w_.data_ptr(), b_.data_ptr(), o_.data_ptr(), S * L, N, K, act, 1, st)
                lib.nova_row_norm(o_.data_ptr(), out_.data_ptr(), gam.data_ptr(), bet.data_ptr(), None, 0, -1, -1, -1, res_.data_ptr(), None, S * L, N, 1e-5, 1, st)
            return run
        ab("out-projection + LayerNorm/residual pass", pair)

        def norm_then_qkv(lib):  # the pair the other way round: the LayerNorm pass, then the GEMM that reads its result (QKV of the next block)
            def run():
                lib.nova_row_norm(o_.data_ptr(), out_.data_ptr(), gam.data_ptr(), bet.data_ptr(), None, 0, -1, -1, -1, res_.data_ptr(), None, S * L, N, 1e-5, 1, st)
                lib.nova_qkv_rope(out_.data_ptr(), w.data_ptr(), b.data_ptr(), rope.data_ptr(), qkv.data_ptr(), S, L, D, heads, 2, 1, st)
            return run
        ab("LayerNorm/residual pass + QKV + RoPE", norm_then_qkv)
        ab("LayerNorm/residual pass alone", lambda lib: lambda: lib.nova_row_norm(o_.data_ptr(), out_.data_ptr(), gam.data_ptr(), bet.data_ptr(), None, 0, -1, -1, -1, res_.data_ptr(), None, S * L, N, 1e-5, 1, st))
    del a_, w_, o_
