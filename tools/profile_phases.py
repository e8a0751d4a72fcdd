#!/usr/bin/env python3
"""Prints where time goes for one frame: per-phase shader cycles of tile_kernel
(thread 0 of every tile, JXLT_FLAG_PROFILE), per-kernel HIP-event times, and the
host stages (D2H fetch, bitstream assembly).  Usage: profile_phases.py [size]"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__  # noqa: E402
import bench  # noqa: E402

PHASES = ["P0 load+XYB", "P1 AQ energy", "P2-3 erosion", "P4 modulations", "P6a 2-block DCTs", "P5a DCT8",
          "P6b entropy", "P7 decision", "P8 quantise", "P9 scan store", "P5b CfL"]
ORDER = [0, 1, 2, 3, 4, 5, 10, 6, 7, 8, 9]  # execution order of the phase slots


def main():
    import torch
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    pkg = __graft_entry__.load_package()
    dev = torch.device("cuda", 0)
    frame = bench.frame_rows_on_device(torch, size, 0, size, 0, dev)
    torch.cuda.synchronize()
    enc = pkg.Encoder(0)
    enc.set_device_image([frame[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=frame)
    for _ in range(2):
        enc.enqueue(1.0, 0)
        enc.synchronize()
    enc.enqueue(1.0, pkg.FLAG_PROFILE)
    enc.synchronize()
    kt = enc.kernel_times()
    ph = np.zeros(16, np.uint64)
    enc._check(enc._L.jxlt_debug_fetch(enc._ctx, 6, ph.ctypes.data, ph.nbytes), "phase fetch")
    ntiles = (size // 64) ** 2
    tot = float(ph.sum())
    print("frame %dx%d, %d tiles; kernel ms: %s" % (size, size, ntiles, {k: round(v, 3) for k, v in kt.items()}))
    for name, c in [(PHASES[i], ph[i]) for i in ORDER]:
        print("  %-20s %9.0f cycles/tile  %5.1f %%" % (name, float(c) / ntiles, 100.0 * float(c) / tot))
    print("  total %.0f cycles/tile (profiled run)" % (tot / ntiles))
    # throughput view: run the kernel truncated after each phase (flags bits 8-11)
    for _ in range(30):  # settle clocks
        enc.enqueue(1.0, 0)
    enc.synchronize()
    res = {}
    for order in (ORDER, list(reversed(ORDER))):
        for i in order:
            ts = []
            for rep in range(5):
                enc.enqueue(1.0, pkg.FLAG_PROFILE | ((i + 1) << 8))
                enc.synchronize()
                ts.append(enc.kernel_times()["tile_kernel"])
            res.setdefault(i, []).append(min(ts))
    prev = 0.0
    for i in ORDER:
        name = PHASES[i]
        t = min(res[i])
        print("  stop after %-22s %7.3f ms  (+%.3f)   [%s]" % (name, t, t - prev, " ".join("%.3f" % v for v in res[i])))
        prev = t
    # host stages
    for rep in range(2):
        t0 = time.perf_counter()
        dp = enc.enqueue(1.0, 0)
        enc.synchronize()
        t1 = time.perf_counter()
        fr = enc.fetch_raw()
        t2 = time.perf_counter()
        blob = enc.assemble(fr, dp, 0)
        t3 = time.perf_counter()
    print("host: device %.2f ms, fetch(D2H) %.2f ms, assemble %.2f ms (%d bytes)" %
          (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), len(blob)))
    # drop-in EncodeFile from pageable host planes (H2D inclusive)
    if size <= 8192:
        host = frame.cpu().numpy()
        for rep in range(2):
            t0 = time.perf_counter()
            jxl = pkg.encode_file(host, 1.0)
            t1 = time.perf_counter()
        print("EncodeFile from host planes (H2D + encode + copy out): %.2f ms = %.0f MP/s (%d bytes)" %
              (1e3 * (t1 - t0), size * size / 1e6 / (t1 - t0), len(jxl)))
        # cjxl_tiny's route: PFM file (tmpfs) -> page-locked read -> payload H2D -> device ingest
        import os, tempfile
        d = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
        path = os.path.join(d, "jxlt_profile_%d.pfm" % os.getpid())
        with open(path, "wb") as f:
            f.write(b"PF\n%d %d\n-1.0\n" % (size, size))
            f.write(np.ascontiguousarray(np.transpose(host, (1, 2, 0))[::-1]).tobytes())
        try:
            for rep in range(2):
                t0 = time.perf_counter()
                jxl2 = pkg.encode_pfm_file(path, 1.0)
                t1 = time.perf_counter()
            print("EncodePFMFile (file in tmpfs -> .jxl bytes, device ingest): %.2f ms = %.0f MP/s (%s)" %
                  (1e3 * (t1 - t0), size * size / 1e6 / (t1 - t0), "same bytes" if jxl2 == jxl else "DIFFERENT BYTES"))
        finally:
            os.unlink(path)
        t0 = time.perf_counter()
        nb = enc.encode_resident(1.0, copy=False)
        print("encode_resident (frame in HBM): %.2f ms = %.0f MP/s" %
              (1e3 * (time.perf_counter() - t0), size * size / 1e6 / (time.perf_counter() - t0)))
    return
    for nt in (1, 8, 32, 64):
        enc.enqueue(1.0, 0)
        fr = enc.fetch_raw()
        t2 = time.perf_counter()
        enc.assemble(fr, dp, nt)
        print("  assemble with %3d threads: %.2f ms" % (nt, 1e3 * (time.perf_counter() - t2)))


if __name__ == "__main__":
    main()
