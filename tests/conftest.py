import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is visible, e.g. `pytest tests/` in the build container
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def abi_calls():
    """{ABI entry point: calls} and {convolution kernel family: launches} made while the test runs: the golden step tests
    assert that the hand-written path -- not a library fallback -- produced the numbers they compare."""
    from vitta_amd import _lib, conv
    _lib.CALL_COUNTS, conv.KERNEL_COUNTS = {}, {}

    class Calls:
        abi, conv_kernels = _lib.CALL_COUNTS, conv.KERNEL_COUNTS

        def assert_tanet_trunk(self, arith=None):
            """every convolution of the step ran through vitta_conv_f32 -- on conv_b3.hip in the default arithmetic"""
            assert self.abi.get("vitta_conv_f32", 0) + self.abi.get("vitta_conv_timed_f32", 0) > 0, self.abi
            assert self.abi.get("vitta_stem_conv7_f32", 0) > 0, self.abi
            if (arith or conv.ARITH) == "b3":
                assert self.conv_kernels.get(_lib.CONV_KERNEL_B3, 0) > 0, self.conv_kernels

        def assert_swin_kernels(self):
            assert self.abi.get("vitta_gemm_nt_f32", 0) + self.abi.get("vitta_gemm_nt_bf16w_f32", 0) > 0, self.abi
            assert any(k.startswith("vitta_wmsa_") and v > 0 for k, v in self.abi.items()), self.abi
            assert any(k.startswith("vitta_ln_") and v > 0 for k, v in self.abi.items()), self.abi

    try:
        yield Calls()
    finally:
        _lib.CALL_COUNTS, conv.KERNEL_COUNTS = None, None
