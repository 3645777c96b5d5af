"""Phase timeline of conv_b3 workgroups (library built with -DB3_TRACE, tools/debug/b3_trace.sh): wave 0 of every workgroup
stamps the shader clock at: 0 entry, 1 requests issued, 2 first stage landed, 3 main loop done, 4 ring drained, 5 partial tile
stored (split K), 6 ticket known, 7 epilogue starts (last arriver: partials re-read), 8 output stores landed."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vitta_amd import _lib, conv as CV  # noqa: E402

dev = torch.device("cuda:0")
so = C.CDLL(os.path.join(os.path.dirname(_lib.__file__), "csrc", "libvitta_hip.so"))
rd = so.vitta_conv_b3_trace_read
rd.argtypes, rd.restype = [C.c_void_p, C.c_int64, C.c_int32], C.c_int
buf = np.zeros(16 * 8192, dtype=np.uint64)


def run(name, n, c, k, h, ksz, stride=1, dgrad=False, flags=0):
    x = torch.randn(c, n * h * h, device=dev)
    w = torch.randn(k, c, ksz, ksz, device=dev) * (c * ksz * ksz) ** -0.5
    geom = CV.Geometry.forward(n, h, h, ksz, stride, ksz // 2)
    y = torch.empty(k, n * geom.hy * geom.wy, device=dev)
    wp = CV.make_pack(CV.pack_fwd(w))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3):
        CV.launch(geom, x, wp, y, c, k)
    rd(None, 0, 1)
    ev[0].record()
    CV.launch(geom, x, wp, y, c, k)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3
    assert rd(buf.ctypes.data, buf.nbytes, 0) == 0
    t = buf.reshape(8192, 16).astype(np.int64)
    used = t[:, 0] > 0
    t = t[used]
    nwg = len(t)
    if os.environ.get("B3_TRACE_RAW"):
        print(t[:3, :13])
    meta = t[:, 12]
    kz, ks, S, xcc = meta & 0xff, (meta >> 8) & 0xff, (meta >> 16) & 0xffff, meta >> 32
    # every XCD has its own shader clock: only differences inside one XCD mean anything
    start, end = np.zeros(nwg), np.zeros(nwg)
    for x in set(xcc.tolist()):
        m = xcc == x
        t0 = t[m, 0].min()
        start[m] = t[m, 0] - t0
        end[m] = np.maximum(t[m, 8], t[m, 6]) - t0
    tick = 1e-3 / 2.4  # us per tick at 2.4 GHz
    last = t[:, 8] > 0
    print(f"== {name}: {nwg} workgroups, ksplit {int(ks.max())}, steps {int(S.max())}, launch {us:.1f} us by events"
          f" (stamps converted at 2.4 GHz)")

    def stat(label, v):
        v = v * tick
        print(f"   {label:46s} mean {v.mean():6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f} us")
    stat("entry -> requests issued", t[:, 1] - t[:, 0])
    if (t[:, 9] > 0).all():
        stat("  entry -> tile / slice indices known", t[:, 9] - t[:, 0])
        stat("  -> epilogue objects, buffer descriptors", t[:, 10] - t[:, 9])
        stat("  -> request offsets, tap validity", t[:, 11] - t[:, 10])
        stat("  -> first NB stages requested", t[:, 13] - t[:, 11])
        stat("  -> epilogue constants / input stream requested", t[:, 1] - t[:, 13])
    stat("requests issued -> first stage landed", t[:, 2] - t[:, 1])
    stat("main loop", t[:, 3] - t[:, 2])
    if (t[:, 14] > 0).any():
        stat("  of it: waiting for the ring + barrier", t[:, 14])
        stat("  of it: issuing the step's requests", t[:, 15])
    stat("ring drained", t[:, 4] - t[:, 3])
    if ks.max() > 1:
        stat("partial tile stored (write-through, drained)", t[:, 5] - t[:, 4])
        stat("ticket round trip", t[:, 6] - t[:, 5])
        stat("last arriver: partials re-read", (t[:, 7] - t[:, 6])[last])
        stat("first finishers' lifetime", (t[:, 6] - t[:, 0])[~last])
    else:
        stat("(no split)", (t[:, 7] - t[:, 4]))
    stat("epilogue until its stores have landed", (t[:, 8] - t[:, 7])[last])
    stat("lifetime of the workgroups that write the output", (t[:, 8] - t[:, 0])[last])


if __name__ == "__main__":
    run("layer2.1.conv2 128->128 3x3 28^2", 16, 128, 128, 28, 3)
    run("layer3.1.conv2 256->256 3x3 14^2", 16, 256, 256, 14, 3)
    run("layer4.1.conv2 512->512 3x3 7^2", 16, 512, 512, 7, 3)
    run("layer1.0.conv3 64->256 1x1 56^2", 16, 64, 256, 56, 1)
    run("layer2.1.conv1 512->128 1x1 28^2", 16, 512, 128, 28, 1)
    run("layer3.1.conv1 1024->256 1x1 14^2", 16, 1024, 256, 14, 1)
    run("layer3.0.conv3 256->1024 1x1 14^2", 16, 256, 1024, 14, 1)
    run("layer4.1.conv1 2048->512 1x1 7^2", 16, 2048, 512, 7, 1)
