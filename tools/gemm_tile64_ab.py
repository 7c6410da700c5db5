"""The denoising loop's GEMMs (bf16, N = K = width, 1100 .. 4000 rows): the 128 x 128 tile (rounds 1-3) against the 64 x 64 tile (round 4)
that the library picks when the 128 tile would give fewer workgroups than CUs. NOVA_GEMM_TILE64 is read when the library loads, so
every measurement is a child process; children alternate. Outputs are hashed: the two tiles must agree bit for bit.
    python tools/gemm_tile64_ab.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import hashlib, os, sys, torch
sys.path.insert(0, sys.argv[1])
from nova_pointcloud_amd import hip
sys.path.insert(0, os.path.join(sys.argv[1], "tools"))
from microbench import timeit
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to("cuda").bfloat16()
for (N, K, act) in ((1024, 1024, 0), (1024, 1024, 2), (768, 768, 2), (1536, 1536, 0)):
    w, bias = rnd(N, K), torch.randn(N, generator=g).cuda()
    for M in (1152, 1408, 1632, 2048, 3264, 3968):
        a = rnd(M, K)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ms = min(timeit(lambda: hip.gemm_bias_act(a, w, bias, act, out=out), iters=50, warm=5) for _ in range(3))
        h = hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
        print("RESULT", N, K, act, M, ms, h)
'''
res = {}
for _ in range(2):
    for mode in ("0", "1"):
        env = dict(os.environ, NOVA_GEMM_TILE64=mode)
        out = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, timeout=300, env=env)
        if "RESULT" not in out.stdout:
            print(out.stderr[-800:])
        for line in out.stdout.splitlines():
            if line.startswith("RESULT"):
                _, N, K, act, M, ms, h = line.split()
                res.setdefault((int(N), int(K), int(act), int(M)), {}).setdefault(mode, []).append((float(ms), h))
print("back-to-back launches from Python (host launch cost included); us per launch, best of 2 processes x 3 rounds")
for (N, K, act, M), d in res.items():
    t0, t1 = min(m for m, _ in d["0"]), min(m for m, _ in d["1"])
    same = len({h for v in d.values() for _, h in v}) == 1
    print(f"N={N:5d} K={K:5d} act={act} M={M:5d}: 128 tile {t0 * 1e3:6.1f} us   automatic (64 tile below one 128-tile per CU) {t1 * 1e3:6.1f} us   ({t0 / t1:.2f}x)   outputs identical: {same}")
