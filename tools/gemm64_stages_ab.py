"""The 64 x 64 tile GEMM of the denoising loop with K-tile rings of different depth (builds with -DNOVA_GEMM64_STAGES=n), one process, interleaved
rounds, the library's per-launch HIP events; outputs compared bit for bit with the first build:
    python tools/gemm64_stages_ab.py lib_a.so lib_b.so ..."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip  # noqa: E402

libs = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, argtypes in hip.SIGNATURES.items():
        if hasattr(lib, name):
            getattr(lib, name).argtypes, getattr(lib, name).restype = argtypes, ctypes.c_int
    libs.append((os.path.basename(path), lib))
SLOT = hip.PROF_SLOTS.index("gemm_small_tile")
n = len(hip.PROF_SLOTS)


def per_launch(lib, fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ms, work, cnt = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_longlong * n)()
    lib.nova_prof_enable(1)
    lib.nova_prof_collect(ms, work, cnt, n)
    for _ in range(iters):
        fn()
    lib.nova_prof_collect(ms, work, cnt, n)
    lib.nova_prof_enable(0)
    return ms[SLOT] / max(cnt[SLOT], 1) * 1e3


g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to("cuda").bfloat16()
st = torch.cuda.current_stream().cuda_stream
for (N, K, act) in ((1024, 1024, 0), (1024, 1024, 2), (3072, 1024, 0), (768, 768, 2), (1536, 1536, 2)):
    w, bias = rnd(N, K), torch.randn(N, generator=g).cuda()
    for M in (512, 1152, 1632, 2048, 3264):
        a = rnd(M, K)
        outs, res = {}, {}
        for _ in range(3):
            for name, lib in libs:
                o = outs.setdefault(name, torch.empty(M, N, dtype=torch.bfloat16, device="cuda"))
                lib.nova_debug_force_gemm_tile(64)
                res.setdefault(name, []).append(per_launch(lib, lambda: lib.nova_gemm_bias_act(a.data_ptr(), w.data_ptr(), bias.data_ptr(), o.data_ptr(), M, N, K, act, 1, st)))
                lib.nova_debug_force_gemm_tile(0)
        same = all(torch.equal(outs[libs[0][0]], v) for v in outs.values())
        print(f"N={N} K={K} act={act} M={M}: " + "  ".join(f"{nm}: {min(v):5.1f} us" for nm, v in res.items()) + f"  identical: {same}", flush=True)
