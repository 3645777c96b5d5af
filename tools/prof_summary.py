"""Summarise a rocprofv3 rocpd database (the default --kernel-trace --stats output of ROCm 7.2) into
a CSV like rocprofv3's kernel_stats: name, calls, total_us, avg_us, pct.  Kernel names are shortened.

    python tools/prof_summary.py gpurun_out/prof_r1/r1_results.db profiles/r1_kernel_stats.csv [last_ms]

The moments kernels are listed per launch geometry ("name [grid=N workgroups]"): the same kernel serves the 29-layer
batched launch of the step (2722 workgroups), the 16x streaming launch and 53 single-layer calibration launches, and
one averaged line would describe none of them.  `--split-all` does that for every kernel.
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


def grid_expr(db):
    cols = [r[1] for r in db.execute("PRAGMA table_info(kernels)")]
    if "grid_x" in cols and "workgroup_x" in cols:
        if "grid_y" in cols and "workgroup_y" in cols and "grid_z" in cols and "workgroup_z" in cols:
            return "((grid_x / workgroup_x) * (grid_y / workgroup_y) * (grid_z / workgroup_z))", cols
        return "(grid_x / workgroup_x)", cols
    if "grid_size_x" in cols and "workgroup_size_x" in cols:
        return "(grid_size_x / workgroup_size_x)", cols
    if "grid_size" in cols and "workgroup_size" in cols:
        return "(grid_size / workgroup_size)", cols
    return None, cols


def main(db_path, out_path, last_ms=None, split_all=False):
    db = sqlite3.connect(db_path)
    gexpr, cols = grid_expr(db)
    print("kernels view columns:", cols)
    t1 = db.execute("select max(end) from kernels").fetchone()[0]
    t0 = db.execute("select min(start) from kernels").fetchone()[0] if last_ms is None else t1 - int(last_ms * 1e6)
    # last_ms: steady state only -- kernels that started in the last `last_ms` milliseconds of the trace (drops
    # model construction, calibration and MIOpen's first-call solver warm-up, which runs naive reference convs)
    raw = list(db.execute(f"select name, {gexpr or '0'}, count(*), sum(duration)/1000.0 from kernels where start >= ? "
                          f"group by name, {gexpr or '0'}", (t0,)))
    tot = sum(r[3] for r in raw) or 1.0
    agg = {}
    for name, grid, calls, total in raw:
        k = short(name)
        if gexpr and (split_all or "moments_" in name):
            k = f"{k} [grid={int(grid)} workgroups]"
        a = agg.setdefault(k, [0, 0.0])
        a[0] += calls
        a[1] += total
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
        for k, (calls, total) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, calls, f"{total:.1f}", f"{total / calls:.2f}", f"{100.0 * total / tot:.2f}"])
    print(f"wrote {out_path}: {len(agg)} kernels, {tot / 1e3:.1f} ms of GPU time")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--split-all"]
    main(args[0], args[1], float(args[2]) if len(args) > 2 else None, split_all="--split-all" in sys.argv)
