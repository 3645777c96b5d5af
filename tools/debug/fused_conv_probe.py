"""Is MIOpen's fused conv+bias+relu (aten::miopen_convolution_relu / _add_relu) as fast as the plain convolution
for the TANet layer1/2 shapes?  (Would let eval-BN be folded into the weights and delete the BN forward pass.)"""
import torch, time
import torch.nn.functional as F
dev = torch.device("cuda:0")
torch.manual_seed(0)


def bench(fn, n=30):
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


shapes = [  # (Cin, Cout, HW, k, name)
    (256, 64, 56, 1, "l1.conv1"), (64, 64, 56, 3, "l1.conv2"), (64, 256, 56, 1, "l1.conv3"),
    (512, 128, 28, 1, "l2.conv1"), (128, 128, 28, 3, "l2.conv2"), (128, 512, 28, 1, "l2.conv3"),
    (1024, 256, 14, 1, "l3.conv1"), (256, 256, 14, 3, "l3.conv2"), (256, 1024, 14, 1, "l3.conv3"),
]
for nfr in (16, 8):
    for cin, cout, hw, k, name in shapes:
        x = torch.randn(nfr, cin, hw, hw, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        bias = torch.randn(cout, device=dev)
        z = torch.randn(nfr, cout, hw, hw, device=dev)
        pad = k // 2
        t_plain = bench(lambda: F.conv2d(x, w, None, 1, pad))
        try:
            t_relu = bench(lambda: torch.ops.aten.miopen_convolution_relu(x, w, bias, [1, 1], [pad, pad], [1, 1], 1))
        except Exception as e:
            t_relu = float("nan"); print("relu err", str(e)[:100])
        try:
            t_add = bench(lambda: torch.ops.aten.miopen_convolution_add_relu(x, w, z, 1.0, bias, [1, 1], [pad, pad], [1, 1], 1))
        except Exception as e:
            t_add = float("nan"); print("add_relu err", str(e)[:100])
        t_bias = bench(lambda: F.conv2d(x, w, bias, 1, pad))
        print(f"{nfr:2d} frames {name:9s} plain {t_plain:7.1f} us | +bias {t_bias:7.1f} | conv_relu {t_relu:7.1f} | conv_add_relu {t_add:7.1f}")
