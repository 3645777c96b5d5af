"""Per-parameter deviation of the 2-rank GPU run (gloo) from the reference's batch-of-two golden, as fractions of the test's
bound (tests/test_gpu_entrypoints.py::test_two_ranks_on_one_gpu_equal_reference_batch_of_two)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import helpers as H
import test_dist_cpu as D

if __name__ == "__main__":
    dev = sys.argv[1] if len(sys.argv) > 1 else "cuda:0"
    g = H.golden("tta3_bz2.npz")
    tmp = tempfile.mkdtemp()
    r0, r1 = D._run(tmp, "full", device=dev)
    for i in range(3):
        k = f"sgd_step{i}_"
        for name in map(str, g["sampled_params"]):
            ref = g[k + f"grad::{name}"]
            err = np.abs(r0[f"step{i}_grad::{name}"] - ref)
            mx = np.abs(ref).max()
            idx = np.unravel_index(err.argmax(), err.shape)
            print(f"step{i} {name:55s} err/max {err.max() / mx:9.2e} noise/max {float(g[k + f'noise_grad::{name}']) / mx:9.2e}"
                  f" n>5e-3: {(err > 5e-3 * mx).sum():5d}/{err.size} at {idx}")
