"""Proxy for splitting the adaptation pass by VIEW (round 6): how long do two independent 1-view x 8-frame adaptation steps take when
their captured graphs are replayed CONCURRENTLY on two streams, against one 2-view x 8-frame step?  (Two adapters on two copies of the
model: no shared state; what a split inside one step could at best reach.)"""
import copy
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
from vitta_amd import data, tta  # noqa: E402


def adapter_for(model, mp, vp, tmp, views, device):
    args = B.make_args(tmp, 224, 8, "adam_affine", device, 8)
    args.n_augmented_views = views
    args.spatiotemp_mean_clean_file, args.spatiotemp_var_clean_file = mp, vp
    ad = tta.ViTTAAdapter(tta.SingleDeviceParallel(model), args)
    ds = data.build_tanet_dataset(args, "val", "tta")
    es = data.build_tanet_dataset(args, "val", "eval")
    x, ev = ds[0][0].unsqueeze(0), es[0][0].unsqueeze(0)
    for _ in range(4):
        ad.set_adapt_mode()
        ad.adapt_step(ad.shape_tta_input(x))
    torch.cuda.synchronize()
    # every adapter's graphs on a capture stream of their own: what is keyed per stream (split-K workspaces, arrival tickets) must not be
    # shared by graphs that are replayed concurrently
    torch.cuda.graph.default_capture_stream = torch.cuda.Stream()
    with torch.cuda.stream(torch.cuda.graph.default_capture_stream):
        ad.set_adapt_mode()
        ad.adapt_step(ad.shape_tta_input(x))  # per-stream tables of this stream exist before the capture
        ad.close_hooks()
        ad.evaluate(ad.shape_eval_input(ev))
        ad.add_hooks_back()
    torch.cuda.synchronize()
    ad.capture_graphs(ad.shape_tta_input(x), ad.shape_eval_input(ev), segmented=False, overlap_eval=False)
    ad.adapt_step(ad.shape_tta_input(x))
    torch.cuda.synchronize()
    return ad, ad.shape_tta_input(x), ad.shape_eval_input(ev)


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def main():
    device = torch.device("cuda:0")
    tmp = tempfile.mkdtemp()
    model, mp, vp = B.build_model_and_stats(tmp, 224, 8, device)
    m1, m2 = copy.deepcopy(model), copy.deepcopy(model)
    m3 = copy.deepcopy(model)
    a2, x2, e2 = adapter_for(model, mp, vp, tmp, 2, device)
    a3, _, e3 = adapter_for(m3, mp, vp, tmp, 2, device)  # (its evaluation graph: a third capture stream)
    b1, xb1, _ = adapter_for(m1, mp, vp, tmp, 1, device)
    b2, xb2, _ = adapter_for(m2, mp, vp, tmp, 1, device)
    s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    res = {}
    res["two_views_one_graph_ms"] = timeit(lambda: a2.adapt_step(x2))
    res["one_view_alone_ms"] = timeit(lambda: b1.adapt_step(xb1))

    def both():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur), s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            b1.adapt_step(xb1)
        with torch.cuda.stream(s2):
            b2.adapt_step(xb2)
        cur.wait_stream(s1), cur.wait_stream(s2)

    res["two_one_view_graphs_concurrently_ms"] = timeit(both)

    def with_eval_2v():
        cur = torch.cuda.current_stream()
        s3.wait_stream(cur)
        with torch.cuda.stream(s3):
            a3.evaluate(e3)
        a2.adapt_step(x2)
        cur.wait_stream(s3)

    def with_eval_split():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur), s2.wait_stream(cur), s3.wait_stream(cur)
        with torch.cuda.stream(s3):
            a3.evaluate(e3)
        with torch.cuda.stream(s1):
            b1.adapt_step(xb1)
        with torch.cuda.stream(s2):
            b2.adapt_step(xb2)
        cur.wait_stream(s1), cur.wait_stream(s2), cur.wait_stream(s3)

    res["two_views_one_graph_plus_eval_stream_ms"] = timeit(with_eval_2v)
    res["two_one_view_graphs_plus_eval_stream_ms"] = timeit(with_eval_split)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
