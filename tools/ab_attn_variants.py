"""A/B of the bf16 / head_dim 64 attention structures of ONE libnova_hip.so in one process, interleaved rounds on random
data (cdna_hip_programming rules 24 / 25): variant 0 = 32x32x16 MFMA (attn.hip), 1 / 2 = 16x16x32 MFMA at 32 / 64 query
rows per wave (attn16.hip).   python tools/ab_attn_variants.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_pointcloud_amd import hip  # noqa: E402
from microbench import timeit  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
lib = hip.load()
dt = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
VARIANTS = [0, 1, 2, 3, 4, 5]
for (S, heads, hd, L) in [(64, 16, 64, 2560), (64, 16, 64, 1537), (64, 16, 64, 768), (64, 12, 64, 1280), (64, 16, 96, 2560), (64, 16, 96, 1537)]:
    D = heads * hd
    qkv = torch.randn(S * L, 3 * D, generator=g)
    qkv[:, :D] *= hd ** -0.5 * 1.4426950408889634  # what the fused QKV epilogue delivers
    qkv = qkv.to("cuda").to(dt)
    outs, res = {}, {v: [] for v in VARIANTS}
    base = qkv.data_ptr()
    for _ in range(rounds):
        for var in (VARIANTS if hd == 64 else [0, 3]):  # head_dim 96: the 32x32x16 kernel against the shipped 16x16x32 form
            o = outs.setdefault(var, torch.empty(S * L, D, dtype=dt, device="cuda"))
            lib.nova_debug_set_attn_variant(var)
            f = lambda: lib.nova_attn_fwd(base, base + 2 * D, base + 4 * D, o.data_ptr(), S, heads, L, L, hd, 3 * D, 3 * D, D,
                                          0.6931471805599453, 1, st)  # scale * log2(e) == 1.0f: q counts as pre-scaled
            res[var].append(timeit(f, iters=6, warm=2))
    lib.nova_debug_set_attn_variant(-1)
    q, k, v = qkv[: 2 * L].float().view(2, L, 3, heads, hd).permute(2, 0, 3, 1, 4)
    ref = torch.nn.functional.scaled_dot_product_attention(q * (hd ** 0.5 / 1.4426950408889634), k, v).transpose(1, 2).reshape(2 * L, D)
    fl = 4.0 * S * heads * L * L * hd
    line = f"hd={hd} heads={heads} L={L}: "
    for var in (VARIANTS if hd == 64 else [0, 3]):
        t = sorted(res[var])
        err = ((outs[var][: 2 * L].float() - ref).abs().max() / ref.abs().max()).item()
        line += f" v{var}: min {t[0]:.3f} med {t[len(t) // 2]:.3f} ms = {fl / t[len(t) // 2] / 1e9:5.0f} TF (err {err:.1e}) |"
    print(line, flush=True)
    del qkv, outs
