"""The clock the chip HOLDS under the encoder's matrix kernels, and what their rate is per cycle.

MI355X lowers its shader clock under load (MI355X_MICROARCH.md, 'DVFS give-back'), so TFLOP/s = FLOP per cycle x the clock held, and
`roofline.frac` (priced at the 2.4 GHz peak) is the product of a per-cycle efficiency and clock / 2.4. The diagnostic build
(-DNOVA_CLOCK: two reads of s_memtime and of the constant 100 MHz s_memrealtime per workgroup, at its start and at its end; nothing inside
the loops; the shipped library executes no stamp) gives the clock as cycles / ticks x 100 MHz, median over workgroups, after >= 2 s of
back-to-back launches. Each kernel runs on random operands (what the workload has) and on zero-filled ones (no switching in the data paths:
the clock the chip would hold if only the instruction stream mattered).

    cd nova_pointcloud_amd/csrc && make &&
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DNOVA_CLOCK -c gemm256.hip -o /tmp/clk_gemm256.o &&
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DNOVA_CLOCK -fno-honor-nans -c attn16.hip -o /tmp/clk_attn16.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 gemm.o /tmp/clk_gemm256.o skinny.o attn.o /tmp/clk_attn16.o attn_bwd.o rowops.o rownorm_bwd.o pointset.o capi.o -o ../../build_exp/libnova_clock.so
    python tools/kernel_clock.py build_exp/libnova_clock.so      -> profiles/r04_kernel_clock.txt
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip  # noqa: E402

lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for name, argtypes in hip.SIGNATURES.items():
    if hasattr(lib, name):
        getattr(lib, name).argtypes, getattr(lib, name).restype = argtypes, ctypes.c_int
for name in ("nova_debug_gemm_clock", "nova_debug_attn_clock"):
    getattr(lib, name).argtypes, getattr(lib, name).restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int

dt = torch.bfloat16
S, L, D, heads = 64, 2560, 1024, 16
SECONDS = float(os.environ.get("CLOCK_SECONDS", "2.0"))
PEAK_PER_GHZ = 2500.0 / 2.4  # TFLOP/s of dense bf16 MFMA per GHz of shader clock (MI355X_MICROARCH.md: 2.5 PFLOP/s at 2.4 GHz)
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream


def fill(shape, kind, scale=0.5):
    if kind == "zeros":
        return torch.zeros(*shape, dtype=dt, device="cuda")
    return (torch.randn(*shape, generator=g) * scale).to("cuda").to(dt)


def measure(tag, launch, flop, reader, nwg):
    launch()
    torch.cuda.synchronize()
    t_end = time.time() + SECONDS
    while time.time() < t_end:  # >= 2 s of back-to-back launches: the clock settles
        for _ in range(20):
            launch()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    buf = np.zeros(2048, dtype=np.int64)
    assert reader(buf.ctypes.data, buf.size) == 0
    rec = buf.reshape(1024, 2)[:nwg]
    rec = rec[rec[:, 1] > 0]
    ghz = np.median(rec[:, 0] / rec[:, 1]) * 0.1
    tf = flop / ms * 1e-9
    print(f"{tag:34s} {ms:7.3f} ms  {tf:7.1f} TFLOP/s = {tf / 2500:5.3f} of the 2.4 GHz peak | clock held {ghz:5.3f} GHz ({ghz / 2.4:5.3f} of 2.4) | "
          f"{tf / (ghz * PEAK_PER_GHZ):5.3f} of the matrix pipe's rate at that clock | workgroup life {np.median(rec[:, 0]):9.0f} cycles", flush=True)


for kind in ("random", "zeros", "random"):
    print(f"--- operands: {kind}", flush=True)
    x, w, b = fill((S * L, D), kind), fill((3 * D, D), kind, D ** -0.5), torch.randn(3 * D, device="cuda")
    rope = torch.rand(2, L, 32, 2, device="cuda")
    qkv = torch.empty(S * L, 3 * D, dtype=dt, device="cuda")
    measure("QKV + RoPE 163840 x 3072 x 1024", lambda: lib.nova_qkv_rope(x.data_ptr(), w.data_ptr(), b.data_ptr(), rope.data_ptr(), qkv.data_ptr(), S, L, D, heads, 2, 1, st),
            2.0 * S * L * 3 * D * D, lib.nova_debug_gemm_clock, 256)
    # attention on the rotated q / k / v just produced (zeros stay zeros: bias aside)
    if kind == "zeros":
        qkv.zero_()
    out = torch.empty(S * L, D, dtype=dt, device="cuda")
    base, es, hd = qkv.data_ptr(), 2, D // heads
    measure("attention 64 x 16 heads x 2560^2 x 64", lambda: lib.nova_attn_fwd(base, base + D * es, base + 2 * D * es, out.data_ptr(), S, heads, L, L, hd, 3 * D, 3 * D, D, float(hd) ** -0.5, 1, st),
            4.0 * S * heads * L * L * hd, lib.nova_debug_attn_clock, 1024)
    for tag, N, K, act in (("out-projection 163840 x 1024 x 1024", 1024, 1024, 0), ("fc1 + GELU 163840 x 4096 x 1024", 4096, 1024, 1), ("fc2 163840 x 1024 x 4096", 1024, 4096, 0)):
        a_, w_, b_, o_ = fill((S * L, K), kind), fill((N, K), kind, K ** -0.5), torch.randn(N, device="cuda"), torch.empty(S * L, N, dtype=dt, device="cuda")
        measure(tag, lambda: lib.nova_gemm_bias_act(a_.data_ptr(), w_.data_ptr(), b_.data_ptr(), o_.data_ptr(), S * L, N, K, act, 1, st),
                2.0 * S * L * N * K, lib.nova_debug_gemm_clock, 256)
        del a_, w_, o_
