"""Workload + summariser for PMC passes over the convolution kernels.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
              --kernel-trace -d out -o conv -- python tools/pmc_conv.py --shapes 256,128,56,1,1 ...
    python tools/pmc_conv.py --summarise out/..._counter_collection.csv out.json

Each shape "C,K,H,k,stride[,tile]" runs forward and data gradient 3 times at 16 frames."""
import csv
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def summarise(path, out_json, trace=None):
    import json
    rows = list(csv.DictReader(open(path)))
    dur = {}
    if trace:
        for r in csv.DictReader(open(trace)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for r in rows:
        name = r["Kernel_Name"]
        if "conv_" not in name or "pack" in name:
            continue
        k = name.split("(")[0].split("::")[-1] + " grid=" + r.get("Grid_Size", "?")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in calls[k] and r["Dispatch_Id"] in dur:
            agg[k]["duration_us"] += dur[r["Dispatch_Id"]]
        calls[k].add(r["Dispatch_Id"])
    out = {}
    for k, c in agg.items():
        n = len(calls[k])
        d = {name: v / n for name, v in c.items()}
        d["launches"] = n
        if d.get("SQ_BUSY_CU_CYCLES"):
            d["mfma_busy_fraction_of_busy_cu_time"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * d["SQ_BUSY_CU_CYCLES"])
        if d.get("GRBM_GUI_ACTIVE") and d.get("duration_us"):
            d["clock_ghz"] = d["GRBM_GUI_ACTIVE"] / d["duration_us"] / 1e3
            if "SQ_VALU_MFMA_BUSY_CYCLES" in d:
                d["mfma_busy_fraction_of_kernel"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * d["GRBM_GUI_ACTIVE"])
        if d.get("SQ_WAVE_CYCLES"):
            for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS",
                       "SQ_ACTIVE_INST_VMEM", "SQ_INSTS_VALU_MFMA_F32"):
                if nm in d:
                    d[nm + "/wave_cycles"] = d[nm] / d["SQ_WAVE_CYCLES"]
        out[k] = d
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
        sys.exit(0)
    import torch
    from vitta_amd import conv as CV
    d = torch.device("cuda:0")
    n = 16
    shapes = [a for a in sys.argv[1:] if not a.startswith("--")]
    for spec in shapes:
        v = [int(t) for t in spec.replace("x", ",").split(",")]
        c, k, h, ksz, s = v[:5]
        tile = (v[5] << 16 | v[6]) if len(v) >= 7 else 0
        pad = ksz // 2
        gf = CV.Geometry.forward(n, h, h, ksz, s, pad)
        x = torch.randn(c, n * h * h, device=d)
        w = torch.randn(k, c, ksz, ksz, device=d) * (c * ksz * ksz) ** -0.5
        wf, wb = CV.pack_fwd(w), CV.pack_bwd(w)
        y = torch.empty(k, n * gf.hy * gf.wy, device=d)
        gx = torch.empty(c, n * h * h, device=d)
        for _ in range(3):
            CV.launch(gf, x, wf, y, c, k, tile=tile)
            if not (s == 2 and ksz == 1):
                for g in CV.Geometry.dgrad(n, h, h, ksz, s, pad):
                    CV.launch(g, y, wb, gx, k, c, tile=tile)
        torch.cuda.synchronize()
        print("ran", spec)
