"""End-to-end effect of the persistent GEMM's start stagger (experiments build): bench.py's workload (d48w1024, 2048 points, batch
32, two lanes), one process, interleaved settings. The stagger spreads the workgroups' epilogue store bursts in time (they otherwise
hit the memory system together: profiles/r03_gemm_store_ablation.txt); with two lanes in flight the skew it adds at the end of a
launch can be filled by the other lane's kernels.   python tools/stagger_e2e.py [cycles,cycles,...] [reps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nova_pointcloud_amd"))
from microbench import use_experiments_lib  # noqa: E402

use_experiments_lib()
import bench  # noqa: E402
from nova_pointcloud_amd import hip  # noqa: E402
from nova_pointcloud_amd.sharding import generate_sharded  # noqa: E402

settings = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,8192,16384,32768,49152").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
width, heads, H, W, B = bench.WORKLOADS["d48w1024_2048pts_b32"]
dev = torch.device("cuda", 0)
pipe = bench.build_pipeline(width, heads, H, W, torch.bfloat16, dev)
prompts = bench.synthetic_prompts(B, dev, torch.bfloat16, seed=1234)
gen = torch.Generator(device=dev).manual_seed(0)
step = lambda: generate_sharded(pipe, prompts, 0, 1, num_inference_steps=64, num_diffusion_steps=25, guidance_scale=5, generator=gen)
step()
torch.cuda.synchronize()
res = {s: [] for s in settings}
for _ in range(reps):
    for s in settings:
        hip.call("nova_debug_force_gemm_tile", 30000 + s // 256)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        res[s].append(time.perf_counter() - t0)
hip.call("nova_debug_force_gemm_tile", 30000)
for s, t in res.items():
    print(f"stagger {s:6d} cycles: " + "  ".join(f"{v * 1e3:.0f} ms" for v in t) + f"   best {B * H * W / min(t):.0f} points/s", flush=True)
