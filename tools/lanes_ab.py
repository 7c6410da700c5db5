"""bench.py's workload under (lanes, persistent-GEMM grid[, walk alternation]) combinations:
python lanes_ab.py [batch] [lanes:grid[:alt],...]"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
batch = sys.argv[1] if len(sys.argv) > 1 else "32"
for combo in (sys.argv[2] if len(sys.argv) > 2 else "1:0,2:0,1:0,2:0").split(","):
    lanes, grid, alt = (tuple(int(v) for v in combo.split(":")) + (1,))[:3]
    env = dict(os.environ, NOVA_LANES=str(lanes), NOVA_GEMM_GRID=str(grid), NOVA_WALK_ALT=str(alt))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--steps", "1", "--warmup", "1", "--batch", batch],
                         env=env, capture_output=True, text=True, timeout=280)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if line:
        j = json.loads(line[-1])
        print(f"lanes={lanes} grid={grid} alt={alt}: {j['value']:.0f} points/s  {j['ms_per_step']:.0f} ms/step", flush=True)
    else:
        print(f"lanes={lanes} grid={grid}: FAILED\n{out.stderr[-800:]}", flush=True)
