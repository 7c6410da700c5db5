"""One ViT block in the fp8 GEMM mode (nova_vit_blocks_forward_fp8: QKV + RoPE, fc1 + GELU with e4m3 output, fc2 on the
block-scaled fp8 MFMA; attention, out-projection and LayerNorms in bf16) at the metric's shapes, for rocprofv3 passes:
    bash tools/pmc_collect.sh gpurun_out/r3/pmc_fp8 pmc_kernels_fp8.py [width heads L S iters]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nova_pointcloud_amd"))
from nova_pointcloud_amd import engine as E  # noqa: E402
from nova_pointcloud_amd import hip  # noqa: E402
from diffnext.models.vision_transformer import Block  # noqa: E402

D, heads, L, S, iters = (int(v) for v in (sys.argv[1:6] + ["1024", "16", "2560", "64", "3"][len(sys.argv) - 1:]))
torch.manual_seed(0)
dt, dev = torch.bfloat16, "cuda"
blk = Block(D, heads).to(dev).to(dt).eval()
hidden = blk.mlp.fc1.out_features
x = (torch.randn(S, L, D, generator=torch.Generator().manual_seed(1)) * 0.7).to(dev).to(dt).reshape(S * L, D).contiguous()
rope = torch.rand(1, L, D // heads // 2, 2, device=dev)
pack, q = E.pack_vit_blocks([blk], dt), E.pack_vit_blocks_fp8([blk])
rows = S * L
e = lambda n, t=dt: torch.empty(rows, n, dtype=t, device=dev)
qkv, a, b, h = e(3 * D), e(D), e(D), e(hidden)
x8, h8 = e(D, torch.uint8), e(hidden, torch.uint8)
xs, hs = torch.empty(rows, dtype=torch.float32, device=dev), torch.empty(rows, dtype=torch.float32, device=dev)
sc = torch.full((1,), 64.0 / 448.0, dtype=torch.float32, device=dev)
am = torch.zeros(1, dtype=torch.int32, device=dev)
with torch.no_grad():
    for _ in range(iters):
        hip.call("nova_vit_blocks_forward_fp8", pack.arr, q.arr, 1, x.data_ptr(), S, L, D, heads, hidden, rope.data_ptr(), 1, qkv.data_ptr(),
                 a.data_ptr(), b.data_ptr(), h.data_ptr(), x8.data_ptr(), xs.data_ptr(), h8.data_ptr(), hs.data_ptr(), sc.data_ptr(),
                 am.data_ptr(), hip.stream_ptr())
torch.cuda.synchronize()
print("pmc_kernels_fp8: done", float(x.float().abs().mean()))
