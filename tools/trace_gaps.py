"""Where one stream's time goes in a rocprofv3 kernel trace: per queue, the busy time, the idle time between consecutive
kernels, and per kernel name the launches, mean duration and mean gap that FOLLOWS such a kernel.

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 bench.py --workload d48w768_1024pts_b8 --steps 1 --warmup 1
    python3 tools/trace_gaps.py out/t_kernel_trace.csv [--from-fraction 0.5]    # second half = the timed step

Written for the launch-latency-bound denoising loop at small batch (thousands of 5-15 us kernels per AR step).
"""
import collections
import csv
import sys


def main(path, from_fraction=0.0):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Queue_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort(key=lambda r: r[1])
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    cut = t0 + (t1 - t0) * from_fraction
    rows = [r for r in rows if r[1] >= cut]
    print(f"{len(rows)} dispatches over {(t1 - cut) / 1e6:.1f} ms")
    by_q = collections.defaultdict(list)
    for r in rows:
        by_q[r[0]].append(r)
    for q, rs in sorted(by_q.items()):
        if len(rs) < 100:
            continue
        busy = sum(r[2] - r[1] for r in rs)
        span = rs[-1][2] - rs[0][1]
        stats = collections.defaultdict(lambda: [0, 0, 0])
        for a, b in zip(rs, rs[1:] + [None]):
            name = a[3].split("(")[0].replace("void nova::", "")[:60]
            s = stats[name]
            s[0] += 1
            s[1] += a[2] - a[1]
            if b is not None:
                s[2] += max(0, b[1] - a[2])
        print(f"queue {q}: {len(rs)} kernels, span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms, idle between kernels {(span - busy) / 1e6:.1f} ms")
        for name, (n, d, g) in sorted(stats.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:14]:
            print(f"    {name:60s} {n:6d} x {d / n / 1e3:7.2f} us  + gap {g / n / 1e3:6.2f} us   = {(d + g) / 1e6:7.1f} ms")


if __name__ == "__main__":
    frac = float(sys.argv[sys.argv.index("--from-fraction") + 1]) if "--from-fraction" in sys.argv else 0.0
    main(sys.argv[1], frac)
