"""The C-ABI boundary without a GPU: libvitta_hip.so builds / loads here (hipcc cross-compiles gfx950), exports every
function include/vitta_hip.h declares, the ctypes binding table names exactly those, the host-only entry points answer,
and the product fails loudly -- no CPU fallback -- when the library is missing or handed host tensors.
No kernel is launched."""
import ctypes
import os
import re

import pytest
import torch

import helpers as H  # noqa: F401  (puts the repo root on sys.path)
from vitta_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vitta_hip.h")


def declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # drop comments
    names = re.findall(r"^\s*(?:const\s+char\s*\*|int64_t|int|size_t|void)\s+(vitta_[a-z0-9_]+)\s*\(", text, flags=re.M)
    assert len(names) == len(set(names)) and len(names) >= 55, len(names)
    return names


@pytest.fixture(scope="module")
def handle():
    path = build.build_lib(verbose=False)  # no-op when the in-tree .so is newer than its sources
    assert path == _lib.LIB_PATH and os.path.exists(path)
    return ctypes.CDLL(path)


def test_library_exports_every_declared_symbol(handle):
    missing = [n for n in declared() if not hasattr(handle, n)]
    assert not missing, missing


def test_binding_table_names_exactly_the_header():
    assert set(_lib.SIGNATURES) == set(declared())


def test_every_entry_point_has_c_linkage_and_plain_types():
    """extern "C", pointers / integers / floats only: no torch or C++ types cross the boundary."""
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)  # declarations only (comments cite torch ops)
    assert 'extern "C"' in text and "torch" not in text.lower() and "std::" not in text and "at::" not in text
    for name, (res, args) in _lib.SIGNATURES.items():
        assert res in (ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_char_p, None), name
        for a in args:
            assert a in (ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t, ctypes.c_int,
                         ctypes.POINTER(ctypes.c_void_p)) or issubclass(a, (ctypes._Pointer, ctypes.Structure)), (name, a)


def test_host_only_entry_points_answer_without_a_gpu():
    L = _lib.lib()
    assert L.vitta_abi_version() == 1
    assert L.vitta_status_string(0).decode().lower().startswith("ok")
    assert "invalid" in L.vitta_status_string(-1).decode().lower()
    # argument validation happens before any device work: null pointers / bad sizes come back as status codes
    assert L.vitta_frames_resample_norm_f32(None, 1, 1, 1, 1, None, None, None, 4, None, None, 4, None, None, 1, 1, 1, 1, None) == -1
    assert L.vitta_scale_add_f32(None, None, None, 1, 4, None, None) == -1
    assert L.vitta_moments_workspace_bytes(16, 1024, 196, 0) > 0
    assert L.vitta_wmsa_supported(392, 32) == 1 and L.vitta_ln_supported(1024) == 1


def test_reciprocal_division_of_the_convolution_kernels_equals_floor_division():
    """conv_common.h FastDiv: every index division of a conv_b3 workgroup is a multiply-high by a host-made reciprocal -- exact for
    every dividend below 2^31 (edge divisors, powers of two and their neighbours, the largest dividends, 10^5 random pairs)."""
    import numpy as np
    f = _lib.lib().vitta_conv_fastdiv_host
    rng = np.random.default_rng(0)
    ds = [1, 2, 3, 5, 7, 9, 49, 196, 784, 3136, 12544, 50176, 2 ** 16 - 1, 2 ** 16, 2 ** 16 + 1, 2 ** 30, 2 ** 31 - 1]
    ds += [2 ** k + e for k in range(1, 30) for e in (-1, 0, 1) if 2 ** k + e >= 1]
    for d in ds:
        ns = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, 2 ** 31 - 1, 2 ** 31 - 2, (2 ** 31 - 1) // d * d, (2 ** 31 - 1) // d * d - 1]
        for n in ns:
            if 0 <= n < 2 ** 31:
                assert f(n, d) == n // d, (n, d)
    n = rng.integers(0, 2 ** 31, 100000)
    d = np.where(rng.random(100000) < 0.5, rng.integers(1, 2 ** 31, 100000), rng.integers(1, 70000, 100000))
    for a, b in zip(n.tolist(), d.tolist()):
        assert f(a, b) == a // b, (a, b)
    assert f(-1, 3) == -1 and f(5, 0) == -1 and f(2 ** 31, 3) == -1


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "vitta_amd", "csrc", "no_such_lib.so"))
    with pytest.raises(_lib.VittaHipError, match="no CPU fallback"):
        _lib.lib()


def test_product_ops_refuse_host_tensors():
    from vitta_amd import frames, ops
    with pytest.raises(_lib.VittaHipError, match="GPU"):
        ops.moments(torch.zeros(2, 4, 3, 3), "bn2d")
    with pytest.raises(_lib.VittaHipError, match="GPU"):
        ops.pred_consis(torch.zeros(1, 2, 5))
    plan = frames.FramePlan([frames.ViewSpec((0, 0, 8, 8), (4, 4))], (4, 4), "cpu", (0.5,) * 3, (0.25,) * 3)
    with pytest.raises(_lib.VittaHipError, match="GPU"):
        frames.resample_normalise(torch.zeros(1, 8, 8, 3, dtype=torch.uint8), plan, 1)


def test_bench_refuses_to_run_without_a_gpu_and_keeps_stdout_clean():
    """bench.py measures the HIP path only: on a box without a GPU it stops with a message on stderr, nothing on stdout
    (the driver reads stdout for the one JSON line), no CPU fallback number."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and out.stdout == "" and "no CPU fallback" in out.stderr
