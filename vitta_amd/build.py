"""Build libvitta_hip.so (hipcc, gfx950 only).

    python -m vitta_amd.build [--force]

hipcc cross-compiles without a GPU; the .so stays in-tree (vitta_amd/csrc/) so it
travels with the repo snapshot to the GPU box.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libvitta_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=True):
    srcs = sources()
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "vitta_hip.h")]
    if not force and not _stale(OUT, deps):
        return OUT
    cmd = [HIPCC] + FLAGS + srcs + ["-o", OUT]
    if verbose:
        print("[vitta_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(OUT)
