"""Summarise a rocprofv3 rocpd database (the default --kernel-trace --stats output of ROCm 7.2) into
a CSV like rocprofv3's kernel_stats: name, calls, total_us, avg_us, pct.  Kernel names are shortened.

    python tools/prof_summary.py gpurun_out/prof_r1/r1_results.db profiles/r1_kernel_stats.csv
"""
import csv
import re
import sqlite3
import sys


def short(name, n=110):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    if name.startswith("Cijk_"):
        m = re.search(r"MT\d+x\d+x\d+", name)
        mi = re.search(r"MI\d+x\d+x\d+", name)
        return f"{name[:22]}..{m.group(0) if m else ''}_{mi.group(0) if mi else ''} (rocBLAS/Tensile fp32 GEMM)"
    name = re.sub(r"<.*", "<...>", name) if len(name) > n else name
    return name[:n]


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    agg = {}
    for name, calls, total, avg, pct in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += calls
        a[1] += total
        a[2] += pct
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
        for k, (calls, total, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, calls, f"{total:.1f}", f"{total / calls:.2f}", f"{pct:.2f}"])
    print(f"wrote {out_path}: {len(agg)} kernels, {sum(a[1] for a in agg.values()) / 1e3:.1f} ms of GPU time")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
