"""Streams with a ROLE: handles that are guaranteed pairwise distinct.

`torch.cuda.Stream()` hands out the 32 streams of a per-device pool round-robin, so in a long-lived process two unrelated requests end up
on the SAME HIP stream.  That is harmless for ordering, but this package keys resources per stream -- split-K workspaces and their
arrival tickets, on-demand weight packs, the meeting counters of the TAM launches -- precisely so that launches which may overlap do
not share them.  Round 6 found the failure mode: late in the full GPU test suite the evaluation side stream of an adapter was the same
pool stream as torch's graph-capture stream; the evaluation graph and the adaptation graph, replayed concurrently, then worked on one
split-K workspace and the seventh video of a `tta_standard` run came out NaN.  Every helper stream of the package therefore comes
from here: one stream per (device, role), distinct from every other role's, from the device's default stream, from the stream current
at the time of the request and from torch's default graph-capture stream (which is pinned now, so that it cannot be created later on
a handle this module has given away)."""
import torch

_roles = {}     # (device index, role) -> stream
_handles = {}   # device index -> set of reserved handles


def _capture_stream():
    gcls = torch.cuda.graphs.graph
    if getattr(gcls, "default_capture_stream", None) is None:  # what torch.cuda.graph() does at its first use
        gcls.default_capture_stream = torch.cuda.Stream()
    return gcls.default_capture_stream


def role(device, name):
    """THE stream of `name` on `device` (created at the first request; every adapter / runner / prefetcher of the process shares it --
    they never run concurrently with themselves, and the per-stream buffers keyed by its handle are reused instead of multiplying)."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, name)
    st = _roles.get(key)
    if st is not None:
        return st
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError(f"vitta_amd.streams: the '{name}' stream must exist before a graph capture starts (run one eager step)")
    taken = _handles.setdefault(idx, set())
    busy = set(taken)
    busy.add(torch.cuda.current_stream(idx).cuda_stream)
    busy.add(torch.cuda.default_stream(idx).cuda_stream)
    cap = _capture_stream()
    if cap.device.index == idx:
        busy.add(cap.cuda_stream)
    for _ in range(128):
        st = torch.cuda.Stream(device=idx)
        if st.cuda_stream not in busy:
            taken.add(st.cuda_stream)
            _roles[key] = st
            return st
    raise RuntimeError("vitta_amd.streams: torch's stream pool has no free handle left")


def roles(device, prefix, count):
    return [role(device, f"{prefix}{i}") for i in range(count)]
