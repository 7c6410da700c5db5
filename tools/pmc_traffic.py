"""Summarise rocprofv3 --pmc counter CSVs for the roofline `traffic` field.

Usage (on the GPU box, one counter per pass as MI355X_MICROARCH.md prescribes):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -o f -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -o w -- python bench.py ...
    python tools/pmc_traffic.py out_f/f_counter_collection.csv out_w/w_counter_collection.csv attn_bf16_hd64

Prints, for the largest-grid launches of the named kernel, the mean HBM bytes per launch with the gfx950
corrections of the guide: FETCH_SIZE is in KiB-units of 1024 B and counts HALF the bytes of wide coalesced
reads (x2); WRITE_SIZE reads exactly for 16-B streaming stores (x1; 8-B stores are uncalibrated).
"""
import csv
import sys
from collections import defaultdict


def load(path, kernel, counter):
    per_dispatch = defaultdict(float)
    grid = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                key = row["Dispatch_Id"]
                per_dispatch[key] += float(row["Counter_Value"])
                grid[key] = int(row.get("Grid_Size", row.get("Grid_Size_X", 0)) or 0)
    if not per_dispatch:
        raise SystemExit(f"no rows for kernel~{kernel} counter={counter} in {path}")
    gmax = max(grid.values())
    vals = [v for k, v in per_dispatch.items() if grid[k] == gmax]
    return sum(vals) / len(vals), len(vals), gmax


if __name__ == "__main__":
    fpath, wpath, kernel = sys.argv[1:4]
    f, nf, g = load(fpath, kernel, "FETCH_SIZE")
    w, nw, _ = load(wpath, kernel, "WRITE_SIZE")
    fetch_bytes, write_bytes = 2.0 * f * 1024.0, w * 1024.0
    print(f"kernel~{kernel} grid={g}: launches {nf}/{nw}  FETCH_SIZE {f:.0f} KiB (x2 -> {fetch_bytes / 1e6:.1f} MB)  "
          f"WRITE_SIZE {w:.0f} KiB ({write_bytes / 1e6:.1f} MB)  total {(fetch_bytes + write_bytes) / 1e6:.1f} MB per launch")
