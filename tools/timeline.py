"""One replayed step as a timeline: every kernel of a window of a rocprofv3 rocpd database with its queue / stream, start offset,
duration and the gap to its predecessor on the same queue -- what the per-kernel averages of prof_summary.py cannot show (which
launches sit on the critical path, where a stream idles, how much two streams overlap).

    python tools/timeline.py <db> <out.csv> [window_ms_from_end] [length_ms]

The window starts `window_ms_from_end` before the last kernel's end (default 40) and is `length_ms` long (default 12: two steps).
"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:70]


def main(db_path, out_path, from_end_ms=40.0, length_ms=12.0):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("PRAGMA table_info(kernels)")]
    print("kernels view columns:", cols)
    qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
    scol = "stream_id" if "stream_id" in cols else ("stream" if "stream" in cols else None)
    gexpr = "(grid_x / workgroup_x)" if "grid_x" in cols and "workgroup_x" in cols else (
        "(grid_size_x / workgroup_size_x)" if "grid_size_x" in cols else "0")
    t1 = db.execute("select max(end) from kernels").fetchone()[0]
    w0 = t1 - int(from_end_ms * 1e6)
    w1 = w0 + int(length_ms * 1e6)
    rows = list(db.execute(f"select name, start, end, {qcol or '0'}, {scol or '0'}, {gexpr} from kernels where start >= ? and start < ? "
                           f"order by start", (w0, w1)))
    last_end = {}
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["t_us", "queue", "stream", "kernel", "workgroups", "dur_us", "gap_us_same_queue"])
        for name, s, e, q, st, g in rows:
            gap = (s - last_end[q]) / 1e3 if q in last_end else ""
            last_end[q] = e
            w.writerow([f"{(s - w0) / 1e3:.2f}", q, st, short(name), int(g or 0), f"{(e - s) / 1e3:.2f}", f"{gap:.2f}" if gap != "" else ""])
    print(f"wrote {out_path}: {len(rows)} kernels in a {length_ms} ms window")


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0], a[1], float(a[2]) if len(a) > 2 else 40.0, float(a[3]) if len(a) > 3 else 12.0)
