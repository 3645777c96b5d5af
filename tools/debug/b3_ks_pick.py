"""Reads gpurun_out/ks_<frames>_<ks>.json (tools/debug/b3_ks_sweep.sh) and prints, per launch of the trunk, the heuristic's time,
the best forced split and the table rows (C, K, taps, M, ks) worth keeping (> 3 % faster than the heuristic)."""
import json
import os

KS = [1, 2, 3, 4, 6, 8, 12, 16]
rows_out, tot_h, tot_b = [], {16: 0.0, 8: 0.0}, {16: 0.0, 8: 0.0}
for frames in (16, 8):
    base = {r["name"]: r for r in json.load(open(f"gpurun_out/ks_{frames}_0.json"))["rows"]}
    forced = {}
    for ks in KS:
        p = f"gpurun_out/ks_{frames}_{ks}.json"
        if os.path.exists(p):
            forced[ks] = {r["name"]: r for r in json.load(open(p))["rows"]}
    for name, r in base.items():
        c, k, h, ksz, s, cnt = r["C"], r["K"], r["H"], r["k"], r["stride"], r["count"]
        ho = (h + 2 * (ksz // 2) - ksz) // s + 1
        for what in ("fwd", "dgrad"):
            if what == "dgrad" and frames == 8:
                continue  # the evaluation pass has no backward
            key = f"{what}_us_b3" if f"{what}_us_b3" in r else f"{what}_us"
            t0 = r[key]
            best_ks, best = 0, t0
            for ks, rr in forced.items():
                t = rr[name][key]
                if t < best:
                    best_ks, best = ks, t
            if what == "fwd":
                desc = (c, k, ksz * ksz, frames * ho * ho)
            elif s == 1:
                desc = (k, c, ksz * ksz, frames * h * h)
            else:
                desc = (k, c, 0 if ksz == 3 else 1, frames * (h // 2) * (h // 2))
            tot_h[frames] += cnt * t0
            gain = 1 - best / t0
            keep = best_ks and gain > 0.03
            tot_b[frames] += cnt * (best if keep else t0)
            print(f"{frames:2d}f {name:22s} {what:5s} heuristic {t0:6.1f} us  best ks={best_ks:2d} {best:6.1f} us  ({100 * gain:4.1f} %)" + ("  *" if keep else ""))
            if keep:
                rows_out.append((desc, best_ks))
print("per pass, heuristic vs tuned (us):", {f: (round(tot_h[f], 1), round(tot_b[f], 1)) for f in tot_h})
seen = {}
for desc, ks in rows_out:
    if desc in seen and seen[desc] != ks:
        print("conflict", desc, seen[desc], ks)
    seen[desc] = ks
for (c, k, t, m), ks in sorted(seen.items()):
    print(f"    {{{c}, {k}, {t}, {m}, {ks}}},")
