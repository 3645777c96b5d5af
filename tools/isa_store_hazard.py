"""Scan the gfx950 assembly of the library's kernels for a wide store whose DATA registers are overwritten by one of the next
instructions: `buffer/global/flat/scratch_store_dwordx3/x4 v[a:b], ...` (or ds_write_b96 / b128) followed within `--window`
instructions by a vector-ALU / load instruction whose destination overlaps v[a:b].  A store of more than 64 bits reads its data
registers after it issues; this compiler inserts no wait state for gfx950, and round 5 met the corruption in a build of conv_b3.hip
(see the comment at its split-K stores).

    python tools/isa_store_hazard.py [--window 2] [file.hip ...]        (default: every vitta_amd/csrc/*.hip)
Prints one line per finding (kernel, store, clobbering instruction, distance) and exits 1 if there is any.
"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# data operand: first for buffer stores, second (behind the address) for global / flat / scratch stores
STORE = re.compile(r"^\s*(buffer_store_dwordx[34])\s+v\[(\d+):(\d+)\]")
GSTORE = re.compile(r"^\s*(global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\s+(?:v\[\d+:\d+\]|v\d+|off),\s*v\[(\d+):(\d+)\]")
DSW = re.compile(r"^\s*(ds_write_b96|ds_write_b128)\s+v\d+,\s*v\[(\d+):(\d+)\]")
DEST = re.compile(r"^\s*(v_\w+|buffer_load\w+|global_load\w+|flat_load\w+|ds_read\w+|scratch_load\w+)\s+(v\[(\d+):(\d+)\]|v(\d+))")


def dest_range(line):
    m = DEST.match(line)
    if not m or m.group(1).startswith(("v_cmp", "v_cmpx", "v_nop")):
        return None
    if m.group(3) is not None:
        return int(m.group(3)), int(m.group(4))
    return int(m.group(5)), int(m.group(5))


# What round 5 met is the MUBUF form (buffer_store_dwordx4 with an SGPR soffset: the rule the Southern Islands .. Vega ISA manuals
# list under "manually inserted wait states").  --all adds global / flat / scratch stores, --lds the LDS writes (which read their data
# at issue): both appear in kernels whose bit-exact tests have always passed, so they are listings, not findings.
INCLUDE_LDS = False
INCLUDE_ALL = False


def scan(asm, window):
    kernel, body, found = None, [], []
    for raw in asm.split("\n"):
        if raw and not raw.startswith(("\t", " ", ".", ";")) and raw.rstrip().endswith(":") or re.match(r"^\w+:\s*;\s*@", raw):
            kernel = raw.split(":")[0]
            body = []
            continue
        line = raw.split(";")[0].rstrip()
        if not line.strip() or line.strip().startswith((".", ";;")) or line.strip().endswith(":"):
            continue
        body.append(line)
        if len(body) > window + 1:
            body.pop(0)
        d = dest_range(line)
        if d is None:
            continue
        for dist, prev in enumerate(reversed(body[:-1]), 1):
            m = STORE.match(prev) or (GSTORE.match(prev) if INCLUDE_ALL else None) or (DSW.match(prev) if INCLUDE_LDS else None)
            if m and not (d[1] < int(m.group(2)) or d[0] > int(m.group(3))):
                found.append((kernel, prev.strip(), line.strip(), dist))
    return found


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", type=int, default=2)
    ap.add_argument("--lds", action="store_true")
    ap.add_argument("--all", action="store_true")
    ap.add_argument("files", nargs="*")
    opt = ap.parse_args()
    global INCLUDE_LDS, INCLUDE_ALL
    INCLUDE_LDS, INCLUDE_ALL = opt.lds, opt.all
    files = opt.files or sorted(glob.glob(os.path.join(ROOT, "vitta_amd", "csrc", "*.hip")))
    bad = 0
    for f in files:
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "k.s")
            r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", f, "-o", out],
                               capture_output=True, text=True)
            if r.returncode != 0:
                print(f"{os.path.basename(f)}: did not compile\n{r.stderr[-400:]}")
                bad += 1
                continue
            hits = scan(open(out).read(), opt.window)
        print(f"{os.path.basename(f)}: {len(hits)} finding(s)")
        for k, st, cl, dist in hits[:20]:
            print(f"    {k[:70]}\n        {st}\n        {cl}   (+{dist})")
        bad += len(hits)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
