"""Tile-group height sweep of the persistent 256-tile GEMM: L2 hit rate, wall and clock per group height, from the
rocprofv3 passes tools/pmc_collect.sh wrote for tools/pmc_gemm_gm.py.   python3 tools/pmc_gm_table.py gpurun_out/r3 2 4 8 16"""
import csv
import glob
import os
import sys
from collections import defaultdict

root, gms = sys.argv[1], [int(v) for v in sys.argv[2:]]
print("| kernel | rows per group | L2 hit | kernel us | clock GHz |")
print("|---|---|---|---|---|")
for gm in gms:
    d = os.path.join(root, f"pmc_gm{gm}")
    ctr, dur = defaultdict(lambda: defaultdict(list)), defaultdict(list)
    for path in glob.glob(os.path.join(d, "*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            ctr[row["Kernel_Name"]][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    for path in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            dur[row["Kernel_Name"]].append((int(row["Dispatch_Id"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3))
    for k in sorted(ctr):
        if "gemm256p" not in k:
            continue
        def per_dispatch(name):
            acc = defaultdict(float)
            for disp, v in ctr[k].get(name, []):
                acc[disp] += v
            vals = [v for _, v in sorted(acc.items())][1:]
            return sum(vals) / len(vals) if vals else None
        hit, miss, gui = per_dispatch("TCC_HIT_sum"), per_dispatch("TCC_MISS_sum"), per_dispatch("GRBM_GUI_ACTIVE")
        us = [v for _, v in sorted(dur.get(k, []))][1:]
        us = sum(us) / len(us) if us else None
        short = "fc1 + GELU" if ", 1>" in k or ",1>" in k else "QKV + RoPE"
        print(f"| {short} | {gm * 256} | {100 * hit / (hit + miss):.1f} % | {us:.1f} | {gui / 8 / us / 1e3:.2f} |" if hit and us and gui else f"| {short} | {gm * 256} | incomplete |")
