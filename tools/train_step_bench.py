"""One training step (forward + backward of Transformer3DModel.train_video) of a random-init BASELINE architecture in bf16
on one MI355X, with the HIP training kernels (attention: csrc/attn16.hip + attn_bwd.hip; LayerNorm family: rowops.hip +
rownorm_bwd.hip; GELU / SiLU: rownorm_bwd.hip act_kernel) switched on and off, same process.

    python3 tools/train_step_bench.py [workload] [batch]        # default d48w1024_2048pts_b32, batch 8
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nova_pointcloud_amd"))
import bench  # noqa: E402
from diffnext.schedulers import FlowMatchEulerDiscreteScheduler  # noqa: E402
from nova_pointcloud_amd import autograd as A  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "d48w1024_2048pts_b32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
width, heads, H, W, _ = bench.WORKLOADS[workload]
model = bench.build_pipeline(width, heads, H, W, torch.bfloat16, "cuda").transformer
model.noise_scheduler = FlowMatchEulerDiscreteScheduler()
model.train()
g = torch.Generator().manual_seed(0)
x = (torch.randn(B, 3, H, W, generator=g) * 0.5).bfloat16().cuda()
prompts = bench.synthetic_prompts(B, "cuda", torch.bfloat16, seed=1)


def step():
    model.zero_grad(set_to_none=True)
    out = model({"x": x.clone(), "prompt": [p.clone() for p in prompts]})
    out["loss"].backward()
    return float(out["loss"].detach())


for use_hip, use_norm, use_act in ((True, True, True), (True, True, False), (True, False, False), (False, False, False)) * 2:
    A._ENABLED, A._NORM_ENABLED, A._ACT_ENABLED = use_hip, use_norm, use_act
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"{workload} batch {B} attention={'HIP' if use_hip else 'torch SDPA'} norms={'HIP fused' if use_norm else 'torch'} activations={'HIP' if use_act else 'torch'}: {ms:.0f} ms per forward+backward, loss {loss:.4f}, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
