"""Workload for the MFMA-utilisation PMC pass: the stage-1 shifted-window attention of Video Swin-B at the C3 shape
(2 views x 16 frames x 224^2 -> 128 windows of 392 tokens, 4 heads of 32), forward + backward, natural token order with
the row map, relative-position table and region mask on chip.  Run under
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
              --kernel-trace -d out -o wmsa -- python tools/pmc_wmsa.py
and summarise with tools/pmc_wmsa.py --summarise out/..._counter_collection.csv"""
import csv
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def summarise(path, out_json):
    import json
    rows = list(csv.DictReader(open(path)))
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for r in rows:
        name = r["Kernel_Name"]
        if "wmsa" not in name:
            continue
        k = name.split("(")[0].split("::")[-1]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    out = {}
    for k, c in agg.items():
        n = len(calls[k])
        d = {name: v / n for name, v in c.items()}
        d["launches"] = n
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CU_CYCLES" in d and d["SQ_BUSY_CU_CYCLES"] > 0:
            # MFMA_BUSY: cycles summed over SIMDs; BUSY_CU: cycles summed over CUs (4 SIMDs each)
            d["mfma_busy_fraction_of_busy_cu_time"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * d["SQ_BUSY_CU_CYCLES"])
        out[k] = d
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2], sys.argv[3])
        sys.exit(0)
    import torch
    from vitta_amd import ops, swin
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, D, H, W, C, nH = 2, 8, 56, 56, 128, 4
    ws, ss = (8, 7, 7), (0, 3, 3)
    n_tok = ws[0] * ws[1] * ws[2]
    attn = swin.WindowAttention3D(C, ws, nH, qkv_bias=True).to(dev)
    rowmap = swin.compute_rowmap(D, H, W, ws, ss, dev)
    region = swin.compute_region(D, H, W, ws, ss, dev)
    qkv = torch.randn(B, D * H * W, 3 * C, device=dev, requires_grad=True)
    for _ in range(6):
        out = ops.WindowAttentionRel.apply(qkv, attn.relative_position_bias_table, attn.relative_position_code[:n_tok],
                                           attn.code_offset, region, attn.scale, nH, rowmap)
        out.square().sum().backward()
        torch.cuda.synchronize()
    print("windows", B * rowmap.shape[0], "tokens", n_tok, "heads", nH)
