"""hipGraph capture probe: fork / join of side streams (a) in the capturing thread, (b) inside an autograd Function's backward
(the engine's worker thread).  python tools/debug/graph_fork_probe.py"""
import sys

import torch
MODE = sys.argv[1] if len(sys.argv) > 1 else "global"
dev = torch.device("cuda:0")
a = torch.randn(1024, 1024, device=dev)
sides = [torch.cuda.Stream(dev) for _ in range(3)]


def body(nfork, rounds):
    main = torch.cuda.current_stream(dev)
    out = []
    for r in range(rounds):
        for st in sides[:nfork]:
            st.wait_stream(main)
        out.append(a @ a)
        for st in sides[:nfork]:
            with torch.cuda.stream(st):
                out.append(a @ a)
        for st in sides[:nfork]:
            main.wait_stream(st)
    return out


class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nfork, rounds):
        ctx.cfg = (nfork, rounds)
        return x * 2

    @staticmethod
    def backward(ctx, g):
        outs = body(*ctx.cfg)
        return g * 2 + outs[-1].sum() * 0, None, None


for nfork, rounds in ((1, 1), (3, 1), (3, 3)):
    body(nfork, rounds)
    x = torch.randn(8, device=dev, requires_grad=True)
    F.apply(x, nfork, rounds).sum().backward()
    torch.cuda.synchronize()
    for where in ("capturing thread", "autograd backward"):
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode=MODE):
                if where == "capturing thread":
                    o = body(nfork, rounds)
                else:
                    x.grad = None
                    F.apply(x, nfork, rounds).sum().backward()
            g.replay()
            torch.cuda.synchronize()
            print("ok  ", where, nfork, rounds)
        except Exception as e:
            print("FAIL", where, nfork, rounds, str(e).splitlines()[0][:90])
            torch.cuda.synchronize()

# (c) an OUTER fork (the evaluation stream of the overlapped step) stays open while inner forks / joins happen in a backward
outer = torch.cuda.Stream(dev)
many = [torch.cuda.Stream(dev) for _ in range(9)]


def nested(n_inner):
    main = torch.cuda.current_stream(dev)
    outer.wait_stream(main)
    with torch.cuda.stream(outer):
        keep = a @ a
    x.grad = None
    sides[:] = many[:3]
    F.apply(x, 3, 1).sum().backward()
    if n_inner > 1:
        sides[:] = many[3:6]
        F.apply(x, 3, 1).sum().backward()
        sides[:] = many[6:9]
        F.apply(x, 3, 1).sum().backward()
    main.wait_stream(outer)
    return keep


for n_inner in (1, 3):
    nested(n_inner)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, capture_error_mode=MODE):
            o = nested(n_inner)
        g.replay()
        torch.cuda.synchronize()
        print("ok   outer fork open,", n_inner, "x 3 inner streams")
    except Exception as e:
        print("FAIL outer fork open,", n_inner, "x 3 inner streams:", str(e).splitlines()[0][:90])
        torch.cuda.synchronize()
