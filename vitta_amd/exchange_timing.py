"""Stream-event timing of the two data-parallel exchanges (norm_stats_utils.py:193,202-204,242-243 pool the batch's clips --
one all-reduce of the packed moments; pred_consistency_utils.py:8,28-30 / the single loss scalar -- one SUM all-reduce of the gradient
arena, in buckets).  bench.py --gpus N switches it on for a few steps behind the timed region so that the first real multi-GPU line says
where the time went: per exchange its bytes and its duration on the launching stream (torch's collectives run on the process group's own
stream, which waits for the launching stream and is joined back into it -- events on the launching stream bracket that for blocking
calls; the bucketed gradient exchange is issued blocking while timing is on, one event pair per bucket).  Nothing is recorded inside a
hipGraph capture (a replayed graph cannot carry events)."""
import contextlib

import torch

RECORDS = None  # None: off.  A list: (kind, bytes, start event, end event) per eagerly launched exchange


def active():
    return RECORDS is not None and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing()


@contextlib.contextmanager
def timed(kind, nbytes):
    if not active():
        yield
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    try:
        yield
    finally:
        e1.record()
        RECORDS.append((kind, int(nbytes), e0, e1))


def summary(n_steps):
    """{kind: {calls_per_step, bytes_per_step, ms_per_step, ms_per_call_max}} of the recorded exchanges (synchronises)."""
    if not RECORDS:
        return {}
    torch.cuda.synchronize()
    out = {}
    for kind, nb, e0, e1 in RECORDS:
        r = out.setdefault(kind, dict(calls=0, bytes=0, ms=0.0, ms_max=0.0))
        ms = e0.elapsed_time(e1)
        r["calls"] += 1
        r["bytes"] += nb
        r["ms"] += ms
        r["ms_max"] = max(r["ms_max"], ms)
    n = max(1, n_steps)
    return {k: dict(calls_per_step=v["calls"] / n, bytes_per_step=v["bytes"] / n, ms_per_step=v["ms"] / n, ms_per_call_max=v["ms_max"])
            for k, v in out.items()}
