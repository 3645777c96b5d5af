"""HBM traffic of the frames kernel (N1) from rocprofv3 PMC passes -- separate runs for FETCH_SIZE and WRITE_SIZE:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_ff -o f -- python tools/pmc_frames.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_fw -o w -- python tools/pmc_frames.py
    python tools/pmc_frames.py --summary <fetch.db> <write.db> profiles/r1p_frames_pmc.json

Workload: 16 TTA clips per launch (32 views x 8 frames of 320x240 -> 224^2), six launches, the first skipped.  A large
untouched buffer is read between launches so that neither the frames nor the tables are cache resident."""
import json
import os
import random
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CLIPS, T, VIEWS, W, H, S = 16, 8, 2, 320, 240, 224
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def specs():
    from vitta_amd import data_video as DV
    from vitta_amd import frames as FR
    random.seed(3)
    out = []
    for _ in range(VIEWS * CLIPS):
        cw, ch, ow, oh = DV.sample_multiscale_crop((W, H), (S, S))
        out.append(FR.ViewSpec((ow, oh, cw, ch), (S, S)))
    return out


def run():
    import numpy as np
    import torch
    from vitta_amd import frames as FR
    dev = torch.device("cuda:0")
    n = VIEWS * CLIPS * T
    frames = torch.from_numpy(np.random.RandomState(0).randint(0, 256, size=(n, H, W, 3)).astype(np.uint8)).to(dev)
    plan = FR.FramePlan(specs(), (S, S), dev, MEAN, STD)
    out = torch.empty(n * 3, S, S, device=dev)
    evict = torch.empty(1 << 28, device=dev)  # 1 GiB
    for _ in range(6):
        evict.add_(1.0)
        FR.resample_normalise(frames, plan, T, out=out)
    torch.cuda.synchronize()


def summary(fetch_db, write_db, out_path):
    def per_launch(db_path, counter):
        db = sqlite3.connect(db_path)
        rows = db.execute("select dispatch_id, sum(value) from counters_collection where counter_name = ? and kernel_name "
                          "like '%frames_resample%' group by dispatch_id order by dispatch_id", (counter,)).fetchall()
        vals = [r[1] for r in rows][1:]
        return sum(vals) / max(1, len(vals)), len(vals)
    f, nf = per_launch(fetch_db, "FETCH_SIZE")
    w, nw = per_launch(write_db, "WRITE_SIZE")
    read_algo = sum(v.box[2] * v.box[3] * 3 for v in specs()) * T
    write_algo = VIEWS * CLIPS * T * 3 * S * S * 4
    res = dict(source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python tools/pmc_frames.py, MI355X",
               units="KiB per launch, averaged over the launches after the first",
               workload=f"{CLIPS} TTA clips per launch ({VIEWS * CLIPS} views x {T} frames of {W}x{H} -> {S}^2)",
               algorithmic_read_bytes=read_algo, algorithmic_write_bytes=write_algo, fetch_size_kib=f, write_size_kib=w,
               hbm_read_bytes_raw=f * 1024.0, hbm_read_bytes_if_halved_counter=2.0 * f * 1024.0, hbm_write_bytes=w * 1024.0,
               write_over_algorithmic=w * 1024.0 / write_algo, read_raw_over_algorithmic=f * 1024.0 / read_algo,
               launches_averaged=[nf, nw],
               note="reads are crop rows fetched in whole cache lines (the crop is a window of each frame row) by unaligned "
                    "dword loads: whether the gfx950 halving of FETCH_SIZE for wide streaming reads applies is not "
                    "established for this access pattern, so both readings are given")
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summary":
        summary(*sys.argv[2:5])
    else:
        run()
