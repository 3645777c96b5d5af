"""What the epilogues cost: the trunk's memory-heavy launches (layer1 / layer2 pointwise convolutions) plain and with the
flags the trunk uses, beside the bytes they move."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitta_amd import conv as CV
from bench_conv import time_it
d = torch.device("cuda:0")
n = 16
for (c, k, h) in [(64, 256, 56), (256, 64, 56), (128, 512, 28), (256, 1024, 14), (1024, 256, 14)]:
    P = n * h * h
    g = CV.Geometry.forward(n, h, h)
    x = torch.randn(c, P, device=d)
    wf = CV.pack_fwd(torch.randn(k, c, 1, 1, device=d) * c ** -0.5)
    y, yr, res = torch.empty(k, P, device=d), torch.empty(k, P, device=d), torch.randn(k, P, device=d)
    bnc = [torch.rand(c, device=d) + 0.5, torch.randn(c, device=d) * 0.1, torch.randn(c, device=d) * 0.1, torch.rand(c, device=d) + 0.5]
    bnk = [torch.rand(k, device=d) + 0.5, torch.randn(k, device=d) * 0.1, torch.randn(k, device=d) * 0.1, torch.rand(k, device=d) + 0.5]
    st = (torch.zeros(k, device=d), torch.zeros(k, device=d), torch.zeros(k, device=d))
    fl = 2.0 * P * c * k
    t0 = time_it(lambda: CV.launch(g, x, wf, y, c, k), 30)
    t1 = time_it(lambda: CV.launch(g, x, wf, y, c, k, flags=CV.CONV_PRO_BN_RELU | CV.CONV_EPI_APPLY | CV.CONV_EPI_RELU | CV.CONV_RES | CV.CONV_STATS,
                                   y_raw=yr, res=res, pro_bn=bnc, epi_bn=bnk, stats=st), 30)
    t2 = time_it(lambda: CV.launch(g, x, wf, y, c, k, flags=CV.CONV_RES, res=res), 30)
    t3 = time_it(lambda: CV.launch(g, x, wf, y, c, k, flags=CV.CONV_BWD_BN | CV.CONV_BWD_RELU, bwd_bn=bnk, bwd_x=res, dgamma=st[0], dbeta=st[1]), 30)
    mb = lambda nin, nout: (c * nin + k * nout) * P * 4 / 1e6
    print(f"C{c} K{k} H{h}: plain {t0:.1f} us ({mb(1,1):.0f} MB) | conv3 fwd epilogue {t1:.1f} us ({mb(1,3):.0f} MB) | +res {t2:.1f} us ({mb(1,2):.0f} MB) | "
          f"bwd-bn epilogue {t3:.1f} us ({mb(1,2):.0f} MB) | mfma floor {fl/157.3e6:.1f} us", flush=True)
