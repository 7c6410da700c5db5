"""bench.py's workload under different numbers of half-batch lanes: python lanes_ab.py [batch] [lanes,lanes,...]"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
batch = sys.argv[1] if len(sys.argv) > 1 else "32"
for lanes in (sys.argv[2] if len(sys.argv) > 2 else "1,2,1,2").split(","):
    env = dict(os.environ, NOVA_LANES=lanes)  # read once at import by nova_pointcloud_amd.engine
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--steps", "1", "--warmup", "1", "--batch", batch],
                         env=env, capture_output=True, text=True, timeout=280)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if line:
        j = json.loads(line[-1])
        print(f"lanes={lanes}: {j['value']:.0f} points/s  {j['ms_per_step']:.0f} ms/step", flush=True)
    else:
        print(f"lanes={lanes}: FAILED\n{out.stderr[-800:]}", flush=True)
