"""profiles/r1_moments_pmc.json from the two rocprofv3 --pmc passes of tools/pmc_moments.py (rocpd databases):
    python tools/pmc_moments_summary.py <fetch.db> <write.db> profiles/r1_moments_pmc.json"""
import json
import sqlite3
import sys


def per_launch(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute("select dispatch_id, grid_size, sum(value) from counters_collection where counter_name = ? and "
                      "kernel_name like '%moments_n%partial%' group by dispatch_id, grid_size order by dispatch_id", (counter,)).fetchall()
    return rows


def main(fetch_db, write_db, out):
    f, w = per_launch(fetch_db, "FETCH_SIZE"), per_launch(write_db, "WRITE_SIZE")
    # tools/pmc_moments.py: 5 launches of the 1-video plan, then 5 of the 16-video plan (same launch geometry, so the
    # two groups are told apart by dispatch order); each group's first launch is skipped
    def groups(rows):
        vals = [r[2] for r in rows]
        half = len(vals) // 2
        return {0: vals[:half], 1: vals[half:]}
    fb, wb = groups(f), groups(w)
    keys = [0, 1]
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python tools/pmc_moments.py, MI355X, round 1 "
                     "(final kernel: layers walked last-first, non-temporal loads)",
           "units": "FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced streaming read "
                    "(MI355X_MICROARCH.md section HBM): read bytes = 2 * FETCH_SIZE * 1024"}
    names = ["in_step_1_video", "streaming_16_videos"]
    algo = [178225152, 2851602432]
    for name, a, k in zip(names, algo, keys):
        fk = sum(fb[k][1:]) / max(1, len(fb[k]) - 1)
        wk = sum(wb[k][1:]) / max(1, len(wb[k]) - 1) if k in wb else float("nan")
        rd, wr = 2.0 * fk * 1024.0, wk * 1024.0
        res[name] = dict(algorithmic_bytes=a, fetch_size_kib=fk, write_size_kib=wk, hbm_read_bytes_corrected=rd,
                         hbm_write_bytes=wr, traffic_over_algorithmic=(rd + wr) / a, launches_averaged=len(fb[k]) - 1)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
