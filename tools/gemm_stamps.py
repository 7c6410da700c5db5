"""Where a tile of the persistent 256x256 GEMM spends its cycles: in-kernel shader-clock stamps of the diagnostic build
(-DNOVA_STAMPS: workgroup 0, one wave of each group, eight points per tile, kept in LDS until the kernel exits) at the four
encoder shapes of the headline workload. The shipped library executes no stamp.

    cd nova_pointcloud_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DNOVA_STAMPS -c gemm256.hip -o /tmp/g.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 gemm.o /tmp/g.o skinny.o attn.o attn16.o attn_bwd.o rowops.o rownorm_bwd.o pointset.o capi.o -o ../../build_exp/libnova_stamps.so
    python tools/gemm_stamps.py build_exp/libnova_stamps.so

Segments per tile (median over the workgroup's tiles but the first and the last; cycles of the shader clock):
  top wait    tile top -> past the wait for K-tile 0 and the barrier(s)
  K-tile 0    first K-tile (accumulators start at the bias)
  K loop      K-tiles 1 .. n-2
  last        last K-tile (RoPE tiles: issues the table rows' LDS-DMA first)
  re-align    the barrier that re-aligns the two wave groups
  epilogue-a  epilogue start -> next tile's prologue issued (RoPE: table rows read from LDS + barrier first)
  epilogue-b  prologue issued -> last store issued
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip  # noqa: E402

lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for name, argtypes in hip.SIGNATURES.items():
    if hasattr(lib, name):
        getattr(lib, name).argtypes, getattr(lib, name).restype = argtypes, ctypes.c_int
lib.nova_debug_gemm_stamps.argtypes, lib.nova_debug_gemm_stamps.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int

dt = torch.bfloat16
S, L, D, heads = 64, 2560, 1024, 16
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to("cuda").to(dt)
st = torch.cuda.current_stream().cuda_stream
TILES, PTS = 64, 8
NAMES = ["top wait", "K-tile 0", "K loop", "last", "re-align", "epilogue-a", "epilogue-b"]


def report(tag, launch, tiles_per_wg):
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    buf = np.zeros(2 * TILES * PTS, dtype=np.uint32)
    assert lib.nova_debug_gemm_stamps(buf.ctypes.data, buf.size) == 0
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5):
        launch()
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / 5
    rec = buf.reshape(2, TILES, PTS).astype(np.int64)
    n = min(tiles_per_wg, TILES)
    line = [f"{tag:22s} {ms:7.3f} ms, {n} tiles per workgroup"]
    for grp in range(2):
        r = rec[grp, 1:n - 1]  # drop the first and the last tile
        seg = np.diff(r, axis=1) & 0xffffffff  # stamps are the low 32 bits of the clock
        med = np.median(seg, axis=0)
        whole = np.median((r[1:, 0] - r[:-1, 0]) & 0xffffffff)
        line.append(f"  group {grp}: tile {whole:8.0f} cycles = " + ", ".join(f"{nm} {v:6.0f}" for nm, v in zip(NAMES, med)))
    print("\n".join(line), flush=True)


x, w, b = rnd(S * L, D), rnd(3 * D, D), torch.randn(3 * D, device="cuda")
rope = torch.rand(2, L, 32, 2, device="cuda")
qkv = torch.empty(S * L, 3 * D, dtype=dt, device="cuda")
report("QKV + RoPE N3072 K1024", lambda: lib.nova_qkv_rope(x.data_ptr(), w.data_ptr(), b.data_ptr(), rope.data_ptr(), qkv.data_ptr(), S, L, D, heads, 2, 1, st), 30)
for tag, N, K, act, tiles in (("proj N1024 K1024", 1024, 1024, 0, 10), ("fc1 + GELU N4096 K1024", 4096, 1024, 1, 40), ("fc2 N1024 K4096", 1024, 4096, 0, 10)):
    a_, w_, b_, o_ = rnd(S * L, K), rnd(N, K), torch.randn(N, device="cuda"), torch.empty(S * L, N, dtype=dt, device="cuda")
    report(tag, lambda: lib.nova_gemm_bias_act(a_.data_ptr(), w_.data_ptr(), b_.data_ptr(), o_.data_ptr(), S * L, N, K, act, 1, st), tiles)
