"""Is a bench workload bound by the host's launch rate or by the GPU? One generation call, timed twice:
host = time until the call returns (everything enqueued, nothing waited for), wall = until the device is idle.
host ~ wall means the host thread is the bottleneck (launch-bound); host << wall means the GPU is.

    python3 tools/host_vs_gpu.py d48w768_1024pts_b8 [lanes]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nova_pointcloud_amd"))
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "d48w768_1024pts_b8"
extra = {"lanes": int(sys.argv[2])} if len(sys.argv) > 2 else {}
width, heads, H, W, B = bench.WORKLOADS[workload]
pipe = bench.build_pipeline(width, heads, H, W, torch.bfloat16, "cuda")
prompts = bench.synthetic_prompts(B, "cuda", torch.bfloat16, seed=1234)
gen = torch.Generator(device="cuda").manual_seed(0)
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe(prompt_embeds=prompts, num_inference_steps=64, num_diffusion_steps=25, guidance_scale=5, generator=gen, output_type="latent", disable_progress_bar=True, **extra)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    from nova_pointcloud_amd import hip
    print(f"{workload} {extra}: host {1e3 * (t1 - t0):.0f} ms, wall {1e3 * (t2 - t0):.0f} ms; decoder graphs (captured, replayed) {hip.graph_stats()}", flush=True)
