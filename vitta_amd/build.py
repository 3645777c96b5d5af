"""Build libvitta_hip.so (hipcc, gfx950 only).

    python -m vitta_amd.build [--force]

hipcc cross-compiles without a GPU; the .so stays in-tree (vitta_amd/csrc/) so it travels with the repo snapshot to the
GPU box.  Every .hip source is compiled to its own object under csrc/.build/ (in parallel) and the objects are linked;
staleness is decided by CONTENT: an object is keyed by the sha256 of its source, of every header it can include and of
the flags, and the library by the sha256 of its objects' keys (stored beside it in libvitta_hip.so.sha256).  A box that
received a prebuilt .so whose key matches its sources compiles nothing; one whose sources differ rebuilds whatever the
file times say.
"""
import glob
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, ".build")
OUT = os.path.join(CSRC, "libvitta_hip.so")
KEYFILE = OUT + ".sha256"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + os.environ.get("VITTA_EXTRA_CFLAGS", "").split()  # (experiments)
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC"]
JOBS = int(os.environ.get("VITTA_BUILD_JOBS", "6"))


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.normpath(os.path.join(HERE, "..", "include", "vitta_hip.h"))]


def _sha(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def library_key():
    """sha256 over every source, header and flag that goes into the library."""
    return _sha(sources() + _headers(), " ".join(CFLAGS + LDFLAGS))


def is_current():
    try:
        return os.path.exists(OUT) and open(KEYFILE).read().strip() == library_key()
    except OSError:
        return False


def _compile(src, hdr_key, verbose):
    key = _sha([src], hdr_key)
    stem = os.path.splitext(os.path.basename(src))[0]
    obj = os.path.join(OBJ, f"{stem}.{key[:16]}.o")
    if not os.path.exists(obj):
        for old in glob.glob(os.path.join(OBJ, f"{stem}.*.o")):
            os.remove(old)
        cmd = [HIPCC] + CFLAGS + ["-c", src, "-o", obj]
        if verbose:
            print("[vitta_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return obj


def build_lib(force=False, verbose=True):
    if not force and is_current():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for old in glob.glob(os.path.join(OBJ, "*.o")):
            os.remove(old)
    hdr_key = _sha(_headers(), " ".join(CFLAGS))
    with ThreadPoolExecutor(max_workers=JOBS) as pool:
        objs = list(pool.map(lambda s: _compile(s, hdr_key, verbose), sources()))
    cmd = [HIPCC] + LDFLAGS + objs + ["-o", OUT]
    if verbose:
        print("[vitta_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(KEYFILE, "w") as f:
        f.write(library_key() + "\n")
    return OUT


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(OUT)
