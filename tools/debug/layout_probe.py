"""NCHW vs channels_last for the ResNet-50 trunk on MI355X (fp32, batch 16 x 224^2): fwd and fwd+bwd."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vitta_amd.resnet import resnet50

dev = torch.device("cuda:0")


def bench(fmt, train):
    torch.manual_seed(0)
    m = resnet50().to(dev).to(memory_format=fmt)
    m.eval()  # BN eval like the TTA step
    x = torch.randn(16, 3, 224, 224, device=dev).to(memory_format=fmt)
    for p in m.parameters():
        p.requires_grad_(train)
    def step():
        if train:
            y = m(x)
            y.sum().backward()
        else:
            with torch.no_grad():
                m(x)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    if not train:
        with torch.cuda.graph(g):
            step()
        run = g.replay
    else:
        run = step
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20


for train in (False, True):
    for fmt in (torch.contiguous_format, torch.channels_last):
        print("train" if train else "infer(graph)", fmt, "%.2f ms" % bench(fmt, train), flush=True)
