"""Round 6: what a fresh 21 MB scratch costs per use (the table-gradient workspace of ops.WindowAttentionRel.backward in EAGER mode)."""
import time, torch
d = torch.device('cuda:0')
n = 5364736
def t(fn, k=50):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6
keep = torch.empty(n, dtype=torch.float32, device=d)
print('fill persistent buffer          %8.1f us' % t(lambda: keep.zero_()))
def fresh():
    w = torch.empty(n, dtype=torch.float32, device=d); w.zero_()
print('empty + fill + free             %8.1f us' % t(fresh))
def fresh_keep():
    w = torch.empty(n, dtype=torch.float32, device=d); w.zero_(); return w
hold = []
print('empty + fill, kept alive        %8.1f us' % t(lambda: hold.append(fresh_keep()), 20))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
w = torch.empty(n, dtype=torch.float32, device=d)
e0.record(); w.zero_(); e1.record(); torch.cuda.synchronize()
print('one fill by events              %8.1f us' % (e0.elapsed_time(e1) * 1e3))
