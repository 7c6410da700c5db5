"""The bf16 / head_dim 64 attention structures at L = 2560 (S = 64, 16 heads), four launches each, for rocprofv3 passes:
    bash tools/pmc_collect.sh gpurun_out/r3/pmc_attn pmc_attn_variants.py
(kernel names tell the structures apart: attn_bf16<...> = 32x32x16, attn_bf16_m16<bf16,64,2|4,false,false|true> = 16x16x32 at 32 /
64 rows per wave without / with the row sums on the matrix pipe, attn_bf16_m16p = software-pipelined)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip  # noqa: E402

lib = hip.load()
S, heads, L, hd = 64, 16, 2560, 64
D = heads * hd
qkv = torch.randn(S * L, 3 * D, generator=torch.Generator().manual_seed(0))
qkv[:, :D] *= hd ** -0.5 * 1.4426950408889634
qkv = qkv.to("cuda").to(torch.bfloat16)
o = torch.empty(S * L, D, dtype=torch.bfloat16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
base = qkv.data_ptr()
for var in (sys.argv[1:] and [int(v) for v in sys.argv[1:]]) or [0, 1, 2, 3, 4, 5]:
    lib.nova_debug_set_attn_variant(var)
    for _ in range(4):
        lib.nova_attn_fwd(base, base + 2 * D, base + 4 * D, o.data_ptr(), S, heads, L, L, hd, 3 * D, 3 * D, D, 0.6931471805599453, 1, st)
    torch.cuda.synchronize()
lib.nova_debug_set_attn_variant(-1)
print("pmc_attn_variants: done", float(o.float().abs().mean()))
