"""profiles/r2_conv_traffic_pmc.json: HBM traffic of the convolution launches of one TTA step, from two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE; separate runs) over `python bench.py --timed-only --no-graph --sequential --steps 6 --warmup 2`:
    python tools/pmc_conv_traffic.py <fetch.db> <write.db> profiles/r2_conv_traffic_pmc.json
Units as tools/pmc_moments_summary.py: the counters are KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read."""
import json
import sqlite3
import sys


def total(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute("select count(distinct dispatch_id), sum(value) from counters_collection where counter_name = ? and "
                      "(kernel_name like '%conv_sk_kernel%' or kernel_name like '%conv_pw_kernel%' or kernel_name like '%conv_igemm_kernel%' "
                      "or kernel_name like '%conv_b3_kernel%' or kernel_name like '%conv_b3w_kernel%')", (counter,)).fetchone()
    return int(rows[0] or 0), float(rows[1] or 0.0)


# algorithmic bytes of the step's 156 convolution launches as bench.py counts them per launch (roofline.algorithmic_bytes_per_step):
# x + the fp32 weights + y + the epilogue's INPUT streams (residual, BatchNorm-backward input, mask), each once; the optional second
# output y_raw is not algorithmic.  (Rounds 2-3 quoted 3.61 GB: x + w + y only.)
ALGO = 4455989248.0


def main(fetch_db, write_db, out, launches_per_step=156, flops_per_step=317529784320.0):
    nf, f = total(fetch_db, "FETCH_SIZE")
    nw, w = total(write_db, "WRITE_SIZE")
    rd_raw, rd, wr = f * 1024.0 / nf, 2.0 * f * 1024.0 / nf, w * 1024.0 / nw
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --timed-only --no-graph "
                     "--sequential --steps 6 --warmup 2 --no-cpu-baseline, MI355X, every convolution dispatch of the run (conv_b3 and the exact-fp32 kernels)",
           "units": "counters are KiB; read bytes = 2 x FETCH_SIZE x 1024 (gfx950 wide-read note, MI355X_MICROARCH.md section HBM; "
                    "the gathered 4-byte loads of the 3x3 launches are outside that calibration)",
           "launches_fetch_pass": nf, "launches_write_pass": nw,
           "hbm_read_bytes_per_launch_as_counted": rd_raw, "hbm_read_bytes_per_launch_corrected": rd,
           "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
           "hbm_bytes_per_step": (rd + wr) * launches_per_step,
           "algorithmic_bytes_per_step": ALGO,
           "traffic_over_algorithmic": (rd + wr) * launches_per_step / ALGO,
           "flop_per_hbm_byte": flops_per_step / ((rd + wr) * launches_per_step)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
