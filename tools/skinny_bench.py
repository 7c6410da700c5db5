"""Small-M GEMM (skinny.hip, 16 / 32 / 64 rows per workgroup) against the 128-tile kernel at the diffusion MLP's per-step
shapes, and the fused modulate -> fc1 -> SiLU launch against row_norm + GEMM.

Launches from Python are host-bound at these sizes, so run it under the profiler and read kernel durations:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -o sk -- python3 tools/skinny_bench.py
    python3 tools/skinny_bench.py --read out/sk_kernel_trace.csv
Every (D, M) point launches each variant 20 times; --read prints the mean duration per variant in launch order.
"""
import csv
import os
import sys

DS = (768, 1024)
MS = (64, 128, 256, 400, 512, 800, 1024, 1600, 2048, 3264)
VARIANTS = (161, 162, 164, 128)
REPS = 20


def run():
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from nova_pointcloud_amd import hip

    dt = torch.bfloat16
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to("cuda").to(dt)
    for D in DS:
        w, bias = rnd(D, D), torch.randn(D, device="cuda")
        for M in MS:
            a, mod = rnd(M, D), rnd(M, 20 * D)
            out = torch.empty(M, D, dtype=dt, device="cuda")
            for tile in VARIANTS:
                hip.call("nova_debug_force_gemm_tile", tile)
                for _ in range(REPS):
                    hip.gemm_bias_act(a, w, bias, 2, out=out)
                torch.cuda.synchronize()
            for tile in (16, 128):  # modulate + fc1: fused launch, then row_norm + GEMM
                hip.call("nova_debug_force_gemm_tile", tile)
                for _ in range(REPS):
                    hip.adaln_fc1(a, mod, D, 2 * D, w, bias, out=out)
                torch.cuda.synchronize()
    hip.call("nova_debug_force_gemm_tile", 0)


def read(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            if "skinny_gemm_kernel" in n or "gemm_kernel<" in n or "row_norm_kernel" in n:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), n))
    rows.sort()
    it = iter(rows)
    take = lambda k: [next(it)[1] for _ in range(k)]
    mean = lambda v: sum(v[2:]) / len(v[2:]) / 1e3  # skip the first two launches of a batch
    for D in DS:
        for M in MS:
            t = {tile: mean(take(REPS)) for tile in VARIANTS}
            fused = mean(take(REPS))
            two = take(2 * REPS)
            print(f"D={D} M={M:5d}: 16 rows {t[161]:6.1f} us  32 rows {t[162]:6.1f}  64 rows {t[164]:6.1f}  128-tile {t[128]:6.1f} | "
                  f"modulate+fc1 fused {fused:6.1f} us  row_norm + GEMM {mean(two[0::2]) + mean(two[1::2]):6.1f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--read":
        read(sys.argv[2])
    else:
        run()
