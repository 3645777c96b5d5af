"""conv1x1 + BN (+res) (+relu) fused GEMM vs F.conv2d + fused BN pass: correctness and time on TANet layer1/2 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn as nn
import torch.nn.functional as F
from vitta_amd import ops
from vitta_amd.fused_bn import bn_act

dev = torch.device("cuda:0")
torch.manual_seed(0)


def bench(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


shapes = [(64, 64, 56, False, "l1.0.conv1"), (256, 64, 56, False, "l1.conv1"), (64, 256, 56, True, "l1.conv3"),
          (256, 128, 56, False, "l2.0.conv1"), (512, 128, 28, False, "l2.conv1"), (128, 512, 28, True, "l2.conv3"),
          (1024, 256, 14, False, "l3.conv1"), (256, 1024, 14, True, "l3.conv3"), (2048, 512, 7, False, "l4.conv1")]
with torch.no_grad():
    for nfr in (16, 8):
        for cin, cout, hw, has_res, name in shapes:
            if (hw * hw) % 4:
                continue
            x = torch.randn(nfr, cin, hw, hw, device=dev)
            conv = nn.Conv2d(cin, cout, 1, bias=False).to(dev)
            bn = nn.BatchNorm2d(cout).to(dev).eval()
            bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 1.5); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
            res = torch.randn(nfr, cout, hw, hw, device=dev) if has_res else None
            ref = bn_act(bn, conv(x), residual=res, relu=True)
            got = ops.conv1x1_bn_act_forward(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, res, True)
            err = (got - ref).abs().max().item() / ref.abs().max().item()
            t_lib = bench(lambda: bn_act(bn, conv(x), residual=res, relu=True))
            t_conv = bench(lambda: conv(x))
            t_own = bench(lambda: ops.conv1x1_bn_act_forward(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                                             bn.eps, res, True))
            gf = 2.0 * cin * cout * hw * hw * nfr / 1e9
            print(f"{nfr:2d} fr {name:10s} err {err:.1e} | conv {t_conv:6.1f} us, conv+bn pass {t_lib:6.1f} us | fused GEMM {t_own:6.1f} us "
                  f"({gf / t_own * 1e3:5.1f} TFLOP/s)")
