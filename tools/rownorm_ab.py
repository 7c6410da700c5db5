"""(Rejected, round 4: the 16-rows-per-wave kernel this script timed measured 188 us against 178 us and was removed; the script needs it rebuilt to run.)
row_norm at the encoder's shape (163840 rows x 1024, bf16, affine + residual): one row per wave (rounds 1-3) against 16 rows per
wave with gamma / beta in registers (round 4). The switch is read when the library loads (NOVA_ROWNORM_ROWS=0 / 1), so every
measurement is a child process; children alternate.  python tools/rownorm_ab.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from nova_pointcloud_amd import hip
sys.path.insert(0, os.path.join(sys.argv[1], "tools"))
from microbench import timeit
rows, D = 64 * 2560, 1024
g = torch.Generator().manual_seed(0)
x = (torch.randn(rows, D, generator=g)).to("cuda").bfloat16()
res = (torch.randn(rows, D, generator=g)).to("cuda").bfloat16()
gamma, beta = torch.randn(D, device="cuda") + 1, torch.randn(D, device="cuda")
out = torch.empty_like(x)
f = lambda: hip.call("nova_row_norm", x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, 0, -1, -1, -1, res.data_ptr(), None,
                     rows, D, 1e-5, 1, hip.stream_ptr())
ts = [timeit(f, iters=20, warm=3) for _ in range(3)]
print("RESULT", min(ts), float(out.float().sum()))
'''
res = {"0": [], "1": []}
for _ in range(3):
    for mode in ("0", "1"):
        env = dict(os.environ, NOVA_ROWNORM_ROWS=mode)
        out = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, timeout=300, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        if not line:
            print(out.stderr[-800:])
            continue
        ms, chk = float(line[-1].split()[1]), line[-1].split()[2]
        res[mode].append((ms, chk))
for mode, name in (("0", "one row per wave"), ("1", "16 rows per wave")):
    ms = [m for m, _ in res[mode]]
    gbs = 3 * 64 * 2560 * 1024 * 2 / min(ms) / 1e6
    print(f"{name:18s}: " + "  ".join(f"{m * 1e3:.1f} us" for m in ms) + f"   best {gbs:.0f} GB/s   checksum {res[mode][0][1]}")
