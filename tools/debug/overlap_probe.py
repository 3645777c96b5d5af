"""Round 6: what the evaluation pass costs beside the adaptation pass under each graph form (same box, same videos):
adapt-only graph | eval-only graph | forked single graph | split graphs | split graphs without the per-step host calls."""
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


def timeit(fn, n=60):
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
    res = {}
    for form in ("plain", "forked", "split"):
        m = copy.deepcopy(model)
        args = B.make_args(tmp, 224, 8, "adam_affine", device, 8)
        args.spatiotemp_mean_clean_file, args.spatiotemp_var_clean_file = mp, vp
        ad = tta.ViTTAAdapter(tta.SingleDeviceParallel(m), args)
        ds = data.build_tanet_dataset(args, "val", "tta")
        es = data.build_tanet_dataset(args, "val", "eval")
        xs = [ad.shape_tta_input(ds[i][0].unsqueeze(0)) for i in range(8)]
        evs = [ad.shape_eval_input(es[i][0].unsqueeze(0)) for i in range(8)]
        for i in range(4):
            ad.set_adapt_mode()
            ad.step(xs[i], evs[i])
        torch.cuda.synchronize()
        if form == "plain":
            ad.capture_graphs(xs[0], evs[0], overlap_eval=False)
            ad.adapt_step(xs[0])
            torch.cuda.synchronize()
            res["adapt_only_graph_ms"] = timeit(lambda: ad.adapt_step(xs[0]))
            res["eval_only_graph_ms"] = timeit(lambda: ad.evaluate(evs[0]))
            k = [0]

            def seq():
                k[0] += 1
                ad.adapt_step(xs[k[0] % 8])
                ad.evaluate(evs[k[0] % 8])
            res["sequential_graphs_ms"] = timeit(seq)
            continue
        ad.capture_graphs(xs[0], evs[0], overlap_eval=True, split=form == "split")
        ad.step(xs[0], evs[0])
        torch.cuda.synchronize()
        k = [0]

        def step():
            k[0] += 1
            ad.step(xs[k[0] % 8], evs[(k[0] - 1) % 8])

        def step_mode():
            ad.set_adapt_mode()
            step()
        res[form + "_step_ms"] = timeit(step)
        res[form + "_step_with_set_adapt_mode_ms"] = timeit(step_mode)
        t0 = time.perf_counter()
        for _ in range(50):
            ad.set_adapt_mode()
        res["set_adapt_mode_host_ms"] = 1e3 * (time.perf_counter() - t0) / 50
        # host cost of issuing one step (no sync in between): how long the calls themselves take
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        res[form + "_step_host_issue_ms"] = 1e3 * (time.perf_counter() - t0) / 20
        torch.cuda.synchronize()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
