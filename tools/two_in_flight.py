#!/usr/bin/env python3
"""Throughput of complete resident encodes of the 16384^2 bench frame with N contexts on one GPU, each driven by
its own host thread (frames in flight overlap: one context's host code construction and codestream download
with the other's kernels).  Usage: two_in_flight.py [size] [frames per context] [contexts ...]"""
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__  # noqa: E402
import bench  # noqa: E402


def main():
    import torch
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    counts = [int(v) for v in sys.argv[3:]] or [1, 2, 3]
    pkg = __graft_entry__.load_package()
    frame = bench.make_frame_on_device(torch, size, 0, torch.device("cuda", 0))
    torch.cuda.synchronize()
    for k in counts:
        encs = [pkg.Encoder(0) for _ in range(k)]
        for e in encs:
            e.set_device_image([frame[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=frame)
            for _ in range(8):
                ref = e.encode_resident(1.0, copy=False).tobytes()
        outs = [None] * k

        def work(i):
            for _ in range(n):
                outs[i] = encs[i].encode_resident(1.0, copy=False)
        threads = [threading.Thread(target=work, args=(i,)) for i in range(k)]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        dt = time.perf_counter() - t0
        same = all(o.tobytes() == ref for o in outs)
        print("%d context(s): %.3f ms per frame = %.1f GP/s, same bytes: %s, tile_kernel %.3f ms" % (
            k, 1e3 * dt / (n * k), size * size * n * k / dt / 1e9, same, encs[0].kernel_times()["tile_kernel"]))
        for e in encs:
            e.close()


if __name__ == "__main__":
    main()
