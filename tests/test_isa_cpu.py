"""The gfx950 assembly of the kernels that issue 16-byte BUFFER stores (split-K partial tiles, hand-over tensors) must not overwrite a
store's data registers with one of the next instructions: the store reads them after it issues and this compiler inserts no wait
state (round 5: a build of conv_b3.hip handed the first data register to the next store's address and a few elements of split tiles
went out corrupted now and then; tools/isa_store_hazard.py, the comment at conv_b3.hip's split-K stores)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("isa_store_hazard", os.path.join(ROOT, "tools", "isa_store_hazard.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_scanner_recognises_the_pattern():
    t = _tool()
    bad = """
_Zkernel:                               ; @_Zkernel
	buffer_store_dwordx4 v[18:21], v34, s[8:11], s7 offen sc1
	v_or_b32_e32 v18, 0x1000, v34
	buffer_store_dwordx4 v[22:25], v18, s[8:11], s7 offen sc1
	s_endpgm
"""
    good = bad.replace("v_or_b32_e32 v18, 0x1000, v34", "v_or_b32_e32 v35, 0x1000, v34").replace("v[22:25], v18", "v[22:25], v35")
    assert len(t.scan(bad, 2)) == 1 and t.scan(bad, 2)[0][3] == 1
    assert t.scan(good, 2) == []


def test_no_wide_buffer_store_is_followed_by_a_write_of_its_data_registers():
    import subprocess
    import tempfile
    t = _tool()
    files = ["conv_b3.hip", "conv.hip", "conv_sk.hip", "head.hip", "tam_branch.hip", "gemm.hip"]
    for f in files:
        src = os.path.join(ROOT, "vitta_amd", "csrc", f)
        assert "buffer_store" in open(src).read(), f
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", out],
                           check=True, capture_output=True)
            hits = t.scan(open(out).read(), 2)
        assert hits == [], (f, hits[:3])
