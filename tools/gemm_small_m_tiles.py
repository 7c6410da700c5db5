"""Plain GEMM (bias / SiLU / GELU epilogues, bf16) at M = 1 .. 1632 rows through each small-M structure, forced: the whole-K kernel of skinny.hip
(its 16 / 32 / 64-row rule of rounds 2-3), gemm.hip's 64 x 64 tile (round 4) and its 128 x 128 tile. Timed with the library's per-launch HIP
events; 64- and 128-tile outputs compared bit for bit. Behind the round-4 selection rule in skinny_row_blocks / launch_gemm.
    python tools/gemm_small_m_tiles.py      -> profiles/r04_gemm_small_m_tiles.txt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip


def timeit(fn, iters=50, warm=5):
    """HIP events around each launch on its stream (the library's profiling slots): excludes the host's launch rate"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    hip.prof_enable(True); hip.prof_collect()
    for _ in range(iters):
        fn()
    ms, _, cnt = hip.prof_collect()["gemm_small_tile"]
    hip.prof_enable(False)
    return ms / max(cnt, 1)

g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to("cuda").bfloat16()
for (N, K, act) in ((1024, 1024, 0), (3072, 1024, 0), (1536, 1536, 2), (4608, 1536, 0), (2304, 768, 0), (1024, 4096, 0), (4096, 1024, 1)):
    w, bias = rnd(N, K), torch.randn(N, generator=g).cuda()
    for M in (1, 8, 16, 32, 48, 64, 96, 128, 256, 512, 1024, 1632):
        a = rnd(M, K); out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        res = {}
        for r in range(3):
            for tile in (16, 64, 128):
                hip.call("nova_debug_force_gemm_tile", tile)
                try:
                    res.setdefault(tile, []).append(timeit(lambda: hip.gemm_bias_act(a, w, bias, act, out=out), iters=50, warm=5))
                except hip.NovaHipError:
                    res.setdefault(tile, []).append(float("nan"))
        ref = None
        outs = {}
        for tile in (128, 64):
            hip.call("nova_debug_force_gemm_tile", tile); o = hip.gemm_bias_act(a, w, bias, act); outs[tile] = o.clone()
        print(f"N={N} K={K} act={act} M={M}: " + "  ".join(f"{'skinny' if t == 16 else t}: {min(v)*1e3:5.1f} us" for t, v in res.items()), " identical:", torch.equal(outs[128], outs[64]), flush=True)
hip.call("nova_debug_force_gemm_tile", 0)
