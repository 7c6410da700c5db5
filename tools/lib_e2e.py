"""bench.py's workload (d48w1024, 2048 points, two lanes; batch 32, or NOVA_E2E_BATCH) on alternative builds of libnova_hip.so, one
subprocess per build and round, interleaved: python tools/lib_e2e.py build_a.so build_b.so ... (timing-only ablation builds give
wrong points). The parent never touches the GPU; every measurement is a fresh child process."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, torch
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nova_pointcloud_amd"))
from nova_pointcloud_amd import hip
hip._LIB_PATH = os.path.abspath(sys.argv[2])
import bench
from nova_pointcloud_amd.sharding import generate_sharded
width, heads, H, W, B = bench.WORKLOADS["d48w1024_2048pts_b32"]
B = int(os.environ.get("NOVA_E2E_BATCH", B))
dev = torch.device("cuda", 0)
pipe = bench.build_pipeline(width, heads, H, W, torch.bfloat16, dev)
prompts = bench.synthetic_prompts(B, dev, torch.bfloat16, seed=1234)
gen = torch.Generator(device=dev).manual_seed(0)
step = lambda: generate_sharded(pipe, prompts, 0, 1, num_inference_steps=64, num_diffusion_steps=25, guidance_scale=5, generator=gen)
step(); torch.cuda.synchronize()
ts = []
for _ in range(2 if B >= 16 else 4):
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("RESULT", min(ts))
'''
libs = sys.argv[1:]
res = {l: [] for l in libs}
for _ in range(2):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", CHILD, ROOT, l], capture_output=True, text=True, timeout=400)
        line = [x for x in out.stdout.splitlines() if x.startswith("RESULT")]
        res[l].append(float(line[-1].split()[1]) if line else float("nan"))
        print(f"  .. {os.path.basename(l)}: {res[l][-1] * 1e3:.0f} ms", file=sys.stderr, flush=True)  # progress line per child
        if not line:
            print(out.stderr[-600:], flush=True)
for l, t in res.items():
    print(f"batch {os.environ.get('NOVA_E2E_BATCH', '32'):>2s}  {os.path.basename(l):40s} " + "  ".join(f"{v * 1e3:.0f} ms" for v in t), flush=True)
