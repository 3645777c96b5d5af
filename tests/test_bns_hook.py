"""stat_reg='BNS' (SURVEY 8f row N3): BNFeatureHook against goldens from the reference; CPU via the oracle
backend, GPU via the HIP moment kernels."""
import pytest
import torch
import torch.nn as nn

import helpers as H
from oracle.oracle_backend import OracleBackend
from vitta_amd.bns_utils import BNFeatureHook

CASES = {"bn2d": (nn.BatchNorm2d, 8, (16, 8, 7, 7)), "bn1d_rows": (nn.BatchNorm1d, 16, (64, 16)),
         "bn1d_nct": (nn.BatchNorm1d, 16, (2, 16, 8))}


def _run(name, reg, device, backend):
    g = H.golden("bns.npz")
    cls, c, shape = CASES[name]
    mod = cls(c).eval()
    gen = torch.Generator().manual_seed(9)
    with torch.no_grad():
        mod.running_mean.copy_(torch.randn(c, generator=gen) * 0.3)
        mod.running_var.copy_(torch.rand(c, generator=gen) + 0.5)
    mod = mod.to(device)
    hook = BNFeatureHook(mod, reg_type=reg, running_manner=True, use_src_stat_in_reg=True, momentum=0.1, backend=backend)
    for step in range(3):
        x = H.channel_feature(shape, 300 + step, 1, offset_scale=0.5).to(device).requires_grad_(True)
        mod(x)
        (gx,) = torch.autograd.grad(hook.r_feature, x)
        key = f"{name}_{reg}_{step}"
        torch.testing.assert_close(hook.r_feature.detach().cpu(), torch.from_numpy(g[key + "_r"]), rtol=2e-5, atol=1e-6)
        torch.testing.assert_close(hook.mean.detach().cpu(), torch.from_numpy(g[key + "_mean"]), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(hook.var.detach().cpu(), torch.from_numpy(g[key + "_var"]), rtol=1e-4, atol=1e-7)
        ref = torch.from_numpy(g[key + "_gx"])
        assert (gx.cpu() - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-9


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("reg", ["l1_loss", "mse_loss", "kld"])
def test_bns_hook_cpu(name, reg):
    _run(name, reg, torch.device("cpu"), OracleBackend())


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("reg", ["l1_loss", "mse_loss", "kld"])
def test_bns_hook_gpu(name, reg):
    _run(name, reg, torch.device("cuda:0"), None)
