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


def summarise(path, out_json, trace=None, match=None):
    """Per (kernel instantiation, grid): the counters averaged over the launches, plus ratios that need no clock assumption
    (everything per wave: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles --
    MI355X_MICROARCH.md, table of per-instruction constants) and one that does (mfma_busy_fraction_of_kernel: 2.4 GHz)."""
    import json
    import re
    rows = list(csv.DictReader(open(path)))
    dur = {}
    if trace:
        for r in csv.DictReader(open(trace)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    waves_of = defaultdict(float)
    for r in rows:
        name = r["Kernel_Name"]
        wg = int(r.get("Workgroup_Size", "256") or 256)
        if match:  # other kernel families (tools/run/pmc_swin.sh): one entry per kernel instantiation, launches of all grids pooled
            m = re.search(match, name)
            if not m:
                continue
            k = m.group(0)
            waves_of[k] += (int(r.get("Grid_Size", "0")) // wg) * (wg // 64) if r["Dispatch_Id"] not in calls[k] else 0
        else:
            if "conv_" not in name or "pack" in name:
                continue
            m = re.search(r"(conv_\w+_kernel<[^>]*>|conv_\w+_kernel)", name)
            k = (m.group(1) if m else name[:60]) + " workgroups=" + str(int(r.get("Grid_Size", "0")) // 256)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in calls[k] and r["Dispatch_Id"] in dur:
            agg[k]["duration_us"] += dur[r["Dispatch_Id"]]
        calls[k].add(r["Dispatch_Id"])
    out = {}
    for k, c in agg.items():
        n = len(calls[k])
        d = {name: v / n for name, v in c.items()}
        d["launches"] = n
        waves = waves_of[k] / n if match else 4.0 * int(k.rsplit("=", 1)[1])
        d["waves_per_launch"] = waves
        if d.get("SQ_WAVE_CYCLES"):
            life = 4.0 * d["SQ_WAVE_CYCLES"] / waves
            d["cycles_per_wave_lifetime"] = life
            for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS",
                       "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
                if nm in d:
                    d[nm + "/wave_cycles"] = d[nm] / d["SQ_WAVE_CYCLES"]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in d:
                # one wave's MFMAs occupy its SIMD's matrix pipe for this share of the wave's lifetime; two workgroups per CU
                # = two waves per SIMD: the pipe is busy about twice that while the waves are alive
                d["mfma_cycles_per_wave"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / waves
                d["mfma_share_of_wave_lifetime"] = d["mfma_cycles_per_wave"] / life
        for nm in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_MFMA"):
            if nm in d:
                d[nm + "_per_wave"] = d[nm] / waves
        if d.get("duration_us") and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            d["mfma_busy_fraction_of_kernel"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * d["duration_us"] * 2400.0)
            d["mfma_busy_fraction_note"] = "matrix-pipe busy cycles / (1024 SIMDs x kernel duration at 2.4 GHz)"
        if d.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_share"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
        out[k] = d
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None, os.environ.get("PMC_MATCH"))
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
