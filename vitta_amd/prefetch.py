"""Input pipeline one batch ahead of the step: host -> device copies on a copy stream while the previous video is being adapted.

The reference feeds the loop from DataLoader workers and a blocking `.cuda()` per batch (corpus/basics.py:431-441, 612-623): decode and
transform overlap with the step through the workers, the upload does not -- it sits in the compute stream in front of the step it
feeds.  `DevicePrefetcher` wraps the loader's iterator: `ahead()` -- called by the loop right after it has issued step i -- pulls batch
i + 1 from the loader, pins it if it is not, and enqueues its copy on a side stream, beside step i on the device; `next()` makes the
CURRENT stream wait for that copy's event and hands out device tensors (what the loop then calls `.to(device)` on is a no-op).  Same
tensors, same order, same StopIteration as the plain iterator (without `ahead()` calls `next()` uploads on demand, still on the copy
stream); on a CPU device it is the plain iterator.
"""
import torch


def _map(obj, fn):
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map(o, fn) for o in obj)
    if isinstance(obj, dict):
        return {k: _map(v, fn) for k, v in obj.items()}
    return obj


class DevicePrefetcher:
    def __init__(self, iterator, device, depth=1):
        self.it = iter(iterator)
        self.device = torch.device(device)
        self.on = self.device.type == "cuda"
        self.depth = max(1, int(depth))
        self.queue = []  # [(batch on the device, copy event, pinned host tensors kept alive until the event)]
        self.done = False
        if self.on:  # (one copy stream per device for every prefetcher: a role stream, distinct from the compute-side roles)
            from . import streams
            self.stream = streams.role(self.device, "copy")
        else:
            self.stream = None
        self.uploads = 0  # batches whose copy ran on the copy stream (tests)

    def _upload(self, batch):
        keep = []

        def up(t):
            if t.is_cuda:
                return t
            if not t.is_pinned():
                t = t.pin_memory()
            keep.append(t)
            return t.to(self.device, non_blocking=True)

        with torch.cuda.stream(self.stream):
            dev = _map(batch, up)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.uploads += 1
        return dev, ev, keep

    def _fill(self):
        while not self.done and len(self.queue) < self.depth:
            try:
                batch = next(self.it)
            except StopIteration:
                self.done = True
                return
            self.queue.append(self._upload(batch))

    def __iter__(self):
        return self

    def __next__(self):
        if not self.on:
            return next(self.it)
        if not self.queue:
            self._fill()
        if not self.queue:
            raise StopIteration
        dev, ev, _keep = self.queue.pop(0)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        _map(dev, lambda t: (t.record_stream(cur), t)[1] if t.is_cuda else t)  # the allocator must not recycle it under the consumer
        return dev

    def ahead(self):
        """Pull the next batch(es) from the loader and start their copies now (call after the current step has been issued)."""
        if self.on:
            self._fill()
