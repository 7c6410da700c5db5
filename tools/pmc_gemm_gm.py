"""fc1 (+GELU) and fused-QKV (+RoPE) GEMMs of config C (M = 64 x 2560, K = 1024) at ONE tile-group height of the persistent
256-tile kernel, for rocprofv3 passes (experiments build: `make -C nova_pointcloud_amd/csrc exp`):
    PMC_PASSES="tcc grbm trace" bash tools/pmc_collect.sh gpurun_out/r3/pmc_gm8 pmc_gemm_gm.py 8"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip  # noqa: E402
from microbench import use_experiments_lib  # noqa: E402

use_experiments_lib()
gm = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dt = torch.bfloat16
S, L, D, heads = 64, 2560, 1024, 16
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to("cuda").to(dt)
x, w1, b1 = rnd(S * L, D), rnd(4 * D, D), torch.randn(4 * D, device="cuda")
wq, bq = rnd(3 * D, D), torch.randn(3 * D, device="cuda")
rope = torch.rand(1, L, 32, 2, device="cuda")
h, qkv = torch.empty(S * L, 4 * D, dtype=dt, device="cuda"), torch.empty(S * L, 3 * D, dtype=dt, device="cuda")
hip.call("nova_debug_force_gemm_tile", 7000 + gm)
for _ in range(4):
    hip.gemm_bias_act(x, w1, b1, 1, out=h)
    hip.qkv_rope(x, wq, bq, rope, S, L, heads, out=qkv)
torch.cuda.synchronize()
hip.call("nova_debug_force_gemm_tile", 7008)
print("pmc_gemm_gm: done", gm)
