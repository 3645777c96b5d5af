"""profiles/r6_moments_rocprof.json from the kernel stats of a PROFILED bench.py run (tools/prof_summary.py CSV): the streaming-size and
in-step launches of moments_nchw_partial_kernel as rocprofv3 timed them, against SURVEY 8d's algorithmic bytes (4 B x hooked elements =
178 225 152 B per video; the streaming launch holds 16 videos' worth).  bench.py reads the file into roofline.moments.rocprof.

    python tools/moments_rocprof.py <kernel_stats.csv> <out.json>
"""
import csv
import json
import sys

PER_VIDEO = 178225152  # 4 B x 44 556 288 hooked elements (TANet-R50, 2 x 8 x 224^2: SURVEY 8a row A1)
HBM_PEAK_GBS = 8000.0


def main(src, dst):
    out = {"kernel": "moments_nchw_partial_kernel", "peak_GBs": HBM_PEAK_GBS, "source_csv": src.split("/")[-1],
           "note": "average launch duration from rocprofv3 --kernel-trace of `python bench.py --no-sgd-all --no-swin` (the bench's own "
                   "figures next to this one come from stream events)"}
    for row in csv.DictReader(open(src)):
        k = row["kernel"]
        if not k.startswith("moments_nchw_partial_kernel"):
            continue
        us = float(row["avg_us"])
        if "grid=5444" in k:
            b = 16 * PER_VIDEO
            out["streaming_16_videos"] = {"bytes": b, "avg_us": us, "calls": int(row["calls"]), "GBs": b / us / 1e3, "frac_of_peak": b / us / 1e3 / HBM_PEAK_GBS}
        elif "grid=2722" in k:
            out["in_step_1_video"] = {"bytes": PER_VIDEO, "avg_us": us, "calls": int(row["calls"]), "GBs": PER_VIDEO / us / 1e3,
                                      "frac_of_peak": PER_VIDEO / us / 1e3 / HBM_PEAK_GBS}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
