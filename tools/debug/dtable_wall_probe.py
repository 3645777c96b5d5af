import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vitta_amd import ops, swin
d = torch.device("cuda:0")
ws = (16, 7, 7); n = 784; nh = 4; c = 128
g = torch.Generator().manual_seed(0)
T = 31 * 13 * 13
table = (torch.randn(T, nh, generator=g) * 0.5).to(d).requires_grad_(True)
table.grad = torch.zeros_like(table)
code, off = swin.relative_position_code(ws); code = code[:n].to(d)
region = torch.randint(0, 4, (4, n), generator=g, dtype=torch.int32).to(d)
qkv = torch.randn(256, n, 3 * c, generator=g).to(d, torch.bfloat16).requires_grad_(True)
gout = torch.randn(256, n, c, generator=g).to(d, torch.bfloat16)
ops.WMSA_BF16 = True
def fb():
    out = ops.WindowAttentionRel.apply(qkv, table, code, off, region, 32 ** -0.5, nh)
    out.backward(gout)
for _ in range(3): fb()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): fb()
torch.cuda.synchronize()
print("wall per fwd+bwd with table gradient: %.1f us" % ((time.perf_counter() - t0) / 20 * 1e6))
