"""rocprofv3 counter CSVs of tools/pmc_collect.sh -> one JSON per round under profiles/ (what bench.py's
`roofline.traffic` and DESIGN.md section 5 cite).

    python3 tools/pmc_summary.py gpurun_out/r2/pmc profiles/r02_pmc.json

Per kernel (mean over the dispatches after the first = warm-up): raw counters, launch duration from the trace pass, and
  hbm_bytes        2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024   (gfx950: FETCH_SIZE reads half the bytes of wide coalesced
                   reads, WRITE_SIZE exact for 16-byte stores - MI355X_MICROARCH.md "HBM")
  clock_ghz        GRBM_GUI_ACTIVE / 8 / duration               (summed over the 8 XCDs)
  mfma_pipe_util   SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8): share of SIMD-cycles with the matrix pipe busy
  valu_active, wait_any, wait_inst, active_any: shares of SQ_WAVE_CYCLES (all quad-cycle counters)
  mfma_valu_coexec SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES
  l2_hit           TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)   (per-XCD L2; MI355X_MICROARCH.md "L2")
  operand_bytes, out_bytes, fetch_over_operands, hbm_over_algorithmic: against the algorithmic bytes of the standard shape
A third argument writes the DESIGN.md table (markdown) generated from the same numbers.
The file carries the sha256 of the kernel sources it was measured on; bench.py ignores it when the sources have changed.
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nova_pointcloud_amd", "csrc")
SIMDS = 1024


def source_hash():
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()


def short(name):
    m = re.match(r"(?:void )?(?:nova::)?([A-Za-z0-9_]+)(<[^(]*>)?", name)
    tmpl = (m.group(2) or "").replace("unsigned short", "bf16").replace("unsigned char", "fp8").replace("nova::f16_t", "f16").replace(" ", "")
    return m.group(1) + tmpl


# One kernel name, two launch shapes: in a ViT block the bias-only 256-tile GEMM is launched twice, out-projection (K = D) then
# fc2 (K = 4 D), always in that order - the dispatches of such a name are told apart by their position in each pass's dispatch order.
SPLIT = {"gemm256c_kernel<bf16,0,false>": ("[proj]", "[fc2]"), "gemm256p_kernel<bf16,0>": ("[proj]", "[fc2]")}


def split_by_order(per_pass):
    """{pass: {kernel: {dispatch: value}}} -> the same with SPLIT names relabelled by dispatch order inside each pass."""
    out = defaultdict(dict)
    for k, d in per_pass.items():
        if k in SPLIT:
            for i, disp in enumerate(sorted(d)):
                out[k + SPLIT[k][i % len(SPLIT[k])]][disp] = d[disp]
        else:
            out[k] = d
    return out


def counters(root):
    acc = defaultdict(lambda: defaultdict(dict))  # kernel -> counter -> dispatch -> value
    for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        one = defaultdict(lambda: defaultdict(dict))  # this pass: counter -> kernel -> dispatch -> value
        with open(path) as f:
            for row in csv.DictReader(f):
                k, c, d = short(row["Kernel_Name"]), row["Counter_Name"], int(row["Dispatch_Id"])
                one[c][k][d] = one[c][k].get(d, 0.0) + float(row["Counter_Value"])
        for c, per_kernel in one.items():
            for k, d in split_by_order(per_kernel).items():
                acc[k][c].update(d)
    return acc


def durations(root):
    out = defaultdict(list)
    for path in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
        one = defaultdict(dict)
        with open(path) as f:
            for row in csv.DictReader(f):
                one[short(row["Kernel_Name"])][int(row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
        for k, d in split_by_order(one).items():
            out[k].extend(sorted(d.items()))
    return out


def mean_skip_first(d):
    vals = [v for _, v in sorted(d.items() if isinstance(d, dict) else d)]
    vals = vals[1:] if len(vals) > 1 else vals
    return sum(vals) / len(vals)


# algorithmic bytes of one launch at the standard shape (S = 64 x L = 2560 rows, D = 1024, 16 heads, bf16): (operands read, output written)
ROWS, D_ = 64 * 2560, 1024
ALGO = {
    "0[proj]": (ROWS * D_ * 2 + D_ * D_ * 2, ROWS * D_ * 2),                # out-projection
    "0[fc2]": (ROWS * 4 * D_ * 2 + 4 * D_ * D_ * 2, ROWS * D_ * 2),         # fc2
    "1": (ROWS * D_ * 2 + 4 * D_ * D_ * 2, ROWS * 4 * D_ * 2),              # fc1 + GELU
    "3": (ROWS * D_ * 2 + 3 * D_ * D_ * 2 + 2560 * 64 * 4, ROWS * 3 * D_ * 2),  # QKV + RoPE (one table)
    "attn": (ROWS * 3 * D_ * 2, ROWS * D_ * 2),
    "row_norm": (2 * ROWS * D_ * 2, ROWS * D_ * 2),
}


def algo_for(k):
    if k.startswith("attn_"):
        return ALGO.get("attn")
    if k.startswith("row_norm"):
        return ALGO.get("row_norm")
    if k.startswith("gemm256"):  # either persistent form: <bf16,EPI[,table-through-LDS]>[label]
        m = re.match(r"gemm256[cp]_kernel<bf16,(\d)(?:,\w+)?>(\[\w+\])?", k)
        return ALGO.get(m.group(1) + (m.group(2) or "")) if m else None
    return ALGO.get(k)


def main():
    args = list(sys.argv[1:])
    shape = "one ViT block, S=64 x L=2560, D=1024, 16 heads, bf16 (tools/pmc_kernels.py)"
    if "--shape" in args:  # another driver / shape: no algorithmic-byte columns (they are written for the standard shape)
        i = args.index("--shape")
        shape = args[i + 1]
        del args[i:i + 2]
        ALGO.clear()
        SPLIT.clear()
    sys.argv = [sys.argv[0]] + args
    root, out_path = sys.argv[1], sys.argv[2]
    acc, dur = counters(root), durations(root)
    kernels = {}
    for k in sorted(acc):
        if not k.startswith(("attn_", "gemm", "row_norm")):
            continue
        raw = {c: mean_skip_first(v) for c, v in acc[k].items()}
        rec = {"counters": {c: round(v, 1) for c, v in sorted(raw.items())}}
        if k in dur:
            rec["duration_us"] = round(mean_skip_first(dur[k]), 2)
        g = lambda c: raw.get(c)
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            rec["hbm_bytes"] = round(2 * g("FETCH_SIZE") * 1024 + g("WRITE_SIZE") * 1024)
            rec["fetch_bytes"], rec["write_bytes"] = round(2 * g("FETCH_SIZE") * 1024), round(g("WRITE_SIZE") * 1024)
        if g("GRBM_GUI_ACTIVE") and rec.get("duration_us"):
            cyc = g("GRBM_GUI_ACTIVE") / 8
            rec["clock_ghz"] = round(cyc / rec["duration_us"] / 1e3, 3)
            if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
                rec["mfma_pipe_util"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (SIMDS * cyc), 4)
        wc = g("SQ_WAVE_CYCLES")
        if wc:
            for name, c in (("valu_active", "SQ_ACTIVE_INST_VALU"),):
                if g(c) is not None:
                    rec[name] = round(g(c) / wc, 4)
        if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_VALU_MFMA_COEXEC_CYCLES") is not None:
            rec["mfma_valu_coexec"] = round(g("SQ_VALU_MFMA_COEXEC_CYCLES") / g("SQ_VALU_MFMA_BUSY_CYCLES"), 4)
        if g("SQ_INSTS_MFMA") and g("SQ_INSTS_VALU") is not None:
            rec["valu_per_mfma"] = round((g("SQ_INSTS_VALU") - g("SQ_INSTS_MFMA")) / g("SQ_INSTS_MFMA"), 2)
        act = g("SQ_ACTIVE_INST_ANY")
        if act is not None and g("SQ_WAIT_ANY") is not None and g("SQ_WAIT_INST_ANY") is not None:
            tot = act + g("SQ_WAIT_ANY") + g("SQ_WAIT_INST_ANY")  # ~ SQ_WAVE_CYCLES of that pass (disjoint buckets)
            rec.update(active_any=round(act / tot, 4), wait_any=round(g("SQ_WAIT_ANY") / tot, 4), wait_inst=round(g("SQ_WAIT_INST_ANY") / tot, 4))
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
            rec["l2_hit"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
        al = algo_for(k)
        if al and "fetch_bytes" in rec:
            rec.update(operand_bytes=al[0], out_bytes=al[1], fetch_over_operands=round(rec["fetch_bytes"] / al[0], 2),
                       hbm_over_algorithmic=round(rec["hbm_bytes"] / (al[0] + al[1]), 2))
        kernels[k] = rec
    doc = {"source_sha256": source_hash(), "shape": shape,
           "units": __doc__.split("Per kernel")[1].strip(), "kernels": kernels}
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    for k, r in kernels.items():
        print(k, {a: b for a, b in r.items() if a != "counters"})
    if len(sys.argv) > 3:  # the DESIGN.md table, generated (never typed by hand)
        rows = ["| kernel | us | clock GHz | matrix pipe busy | VALU per MFMA | active / issue-stall / wait | L2 hit | fabric fetch / operands | HBM bytes / algorithmic |",
                "|---|---|---|---|---|---|---|---|---|"]
        pc = lambda v: "-" if v is None else f"{100 * v:.1f} %"
        for k, r in kernels.items():
            shares = "-" if "active_any" not in r else f"{100 * r['active_any']:.0f} / {100 * r['wait_inst']:.0f} / {100 * r['wait_any']:.0f} %"
            rows.append(f"| `{k}` | {r.get('duration_us', '-')} | {r.get('clock_ghz', '-')} | {pc(r.get('mfma_pipe_util'))} | {r.get('valu_per_mfma', 'no MFMA') if r.get('mfma_pipe_util') else 'no MFMA'} | {shares} | "
                        f"{pc(r.get('l2_hit'))} | {r.get('fetch_over_operands', '-')} | {r.get('hbm_over_algorithmic', '-')} |")
        open(sys.argv[3], "w").write("\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
