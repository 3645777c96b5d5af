"""Summarise a rocprofv3 rocpd database (the default --kernel-trace --stats output of ROCm 7.2) into
a CSV like rocprofv3's kernel_stats: name, calls, total_us, avg_us, pct.  Kernel names are shortened.

    python tools/prof_summary.py gpurun_out/prof_r1/r1_results.db profiles/r1_kernel_stats.csv [last_ms]
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


def main(db_path, out_path, last_ms=None):
    db = sqlite3.connect(db_path)
    if last_ms is None:
        rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    else:
        # steady state only: kernels that started in the last `last_ms` milliseconds of the trace (drops model
        # construction, calibration and MIOpen's first-call solver warm-up, which runs naive reference convs)
        t1 = db.execute("select max(end) from kernels").fetchone()[0]
        raw = list(db.execute("select name, count(*), sum(duration)/1000.0 from kernels where start >= ? group by name",
                              (t1 - int(last_ms * 1e6),)))
        tot = sum(r[2] for r in raw) or 1.0
        rows = [(n, c, d, d / c, 100.0 * d / tot) for n, c, d in raw]
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
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else None)
