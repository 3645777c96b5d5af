"""Measured GEMM selections for the library GEMMs of the Video Swin-B step (qkv / proj / MLP, 60 % of the step).

PyTorch-ROCm routes `F.linear` to hipBLASLt with the library's default heuristic; `torch.cuda.tunable` (TunableOp) can
instead time every hipBLASLt / rocBLAS solution for a GEMM shape once and remember the fastest.  The table committed
here (`tunableop_swin_b_c3_mi355x0.csv`) is that search run ON an MI355X over the 60 GEMM shapes of one per-video
iteration at C3 (2 views x 16 frames x 224^2, forward, input-gradient, evaluation forward) with

    PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunableop_swin.csv \\
        PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 python tools/bench_swin.py --sequential --no-graph

`enable_tuned_gemms()` loads it with tuning OFF: shapes in the table take the recorded solution, every other shape
the library default; nothing is searched at run time.  Same arithmetic (fp32 MFMA GEMMs), different tile choice:
results move by round-off only.  The table is valid for the library versions recorded in its header (TunableOp checks
them and ignores the file otherwise).  Measured: 24.8 -> 21.3 ms per video (sequential), 23.1 -> 21.3 (overlapped).

OFF BY DEFAULT everywhere (`bench.py --arch swin --tuned-gemms`, `tools/bench_swin.py --tuned-gemms`, `--tuned_gemms` of
the entry points): of seven runs with TunableOp enabled on the round's GPU boxes, three did not finish within their
250-300 s limit (the other four took 12 s and produced the figures above).  A watchdog run
(`tools/debug/stall_trace.py`) places the stall on the DEVICE: the host sits in `torch.cuda.synchronize()` after the first
replay of the captured step -- the graph never completes.  Suspected, not proven: some of the selected solutions
coordinate their workgroups through flags in a workspace (split-K / stream-K style) and do not tolerate the replayed
graph's concurrency (two branches issuing GEMMs that share the library workspace).  Until that is settled the table is
evidence of what the library's default heuristic leaves on the table for these shapes (8-14 % of the step), not a
shipped setting.
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
TABLES = {"swin_b_c3": os.path.join(HERE, "tunableop_swin_b_c3_mi355x0.csv")}


def enable_tuned_gemms(table="swin_b_c3"):
    """Returns True when the table was accepted.  GPU only; no effect (False) on the CPU."""
    import torch
    if not torch.cuda.is_available():
        return False
    import torch.cuda.tunable as T
    T.enable(True)
    T.tuning_enable(False)
    ok = bool(T.read_file(TABLES[table]))
    if not ok:
        T.enable(False)
    return ok
