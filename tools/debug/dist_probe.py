import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from vitta_amd import conv as CV
d = torch.device('cuda:0')
shapes = [(4, 64, 64, 56, 1), (2, 256, 64, 56, 1), (16, 64, 256, 56, 1), (8, 512, 128, 28, 1), (16, 1024, 256, 14, 1), (16, 2048, 512, 7, 1), (8, 512, 128, 28, 1), (16, 1024, 256, 14, 1)]
for (n, c, k, h, ksz) in shapes:
    g = torch.Generator().manual_seed(n * 1000 + c + k + h + ksz)
    x = torch.randn(n, c, h, h, generator=g)
    w = torch.randn(k, c, ksz, ksz, generator=g) * c ** -0.5
    ref = F.conv2d(x.double(), w.double())
    geom = CV.Geometry.forward(n, h, h, ksz, 1, 0)
    y = torch.full((k, n * h * h), float('nan'), device=d)
    CV.launch(geom, CV.to_cm(x.to(d)), CV.pack_fwd(w.to(d)), y, c, k)
    torch.cuda.synchronize()
    ws = CV.workspace(d) if hasattr(CV, 'workspace') else None
    got = y.cpu().double()
    r = ref.permute(1, 0, 2, 3).reshape(k, -1)
    err = (got - r).abs()
    bad = err > 1e-3
    M = err.shape[1]
    rows = []
    for mt in range((M + 127) // 128):
        for rb in range(4):
            lo, hi = mt * 128 + rb * 32, min(M, mt * 128 + rb * 32 + 32)
            if lo >= M: continue
            for nt in range(k // 64):
                b = bad[nt * 64:(nt + 1) * 64, lo:hi].float().mean().item()
                if b > 0: rows.append((mt, rb, nt, round(b, 2)))
    cnt = None
    try:
        wsb = ws if torch.is_tensor(ws) else ws[0]
        cnt = wsb[:65536].view(torch.int32)
        nz = (cnt != 0).nonzero().flatten()[:10].tolist()
        cnt = (int((cnt != 0).sum()), nz, [hex(int(cnt[i])) for i in nz])
    except Exception as e:
        cnt = repr(e)
    print((n, c, k, h), 'max err %.2e' % err.max().item(), 'bad blocks', len(rows), rows[:12], 'counters nonzero:', cnt, flush=True)
