#!/usr/bin/env python3
"""What the one-frame-over-N-participants protocol costs per frame, measured on ONE GPU: the 16384^2 bench frame
through jxlt_multi_encoder_* with N device contexts on GPU 0 (threads of one process: the contexts' kernels share
the GPU without the time-slicing that separate processes suffer) against the single-context encode.  The GPU's work
is the same frame in all cases, so the difference is hand-overs, waits and the serial host stage.
Usage: shard_overhead.py [size] [frames]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__  # noqa: E402
import bench  # noqa: E402


def main():
    import torch
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    pkg = __graft_entry__.load_package()
    t = bench.frame_rows_on_device(torch, size, 0, size, 0, torch.device("cuda", 0))
    torch.cuda.synchronize()
    enc = pkg.Encoder(0)
    enc.set_device_image([t[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=t)
    for _ in range(5):
        single = enc.encode_resident(1.0, copy=False)
    t0 = time.perf_counter()
    for _ in range(frames):
        single = enc.encode_resident(1.0, copy=False)
    base = (time.perf_counter() - t0) / frames
    single = single.tobytes()
    print("1 context            %.3f ms per frame" % (base * 1e3))
    for n in (2, 4, 8):
        me = pkg.MultiEncoder([0] * n)
        for slab in range(n):
            x0, y0, x1, y1 = pkg.shard_rect(size, size, n, slab)
            if y1 > y0:
                me.set_device_slab(slab, [t[c, y0:, x0:].data_ptr() for c in range(3)], size * 4, x1 - x0, y1 - y0, keepalive=t)
        for _ in range(5):
            out = me.encode_resident(size, size, 1.0)
        t0 = time.perf_counter()
        for _ in range(frames):
            out = me.encode_resident(size, size, 1.0)
        dt = (time.perf_counter() - t0) / frames
        print("%d contexts (threads)  %.3f ms per frame  (+%.3f ms)  same bytes: %s" %
              (n, dt * 1e3, (dt - base) * 1e3, out.tobytes() == single))
        me.close()


if __name__ == "__main__":
    main()
