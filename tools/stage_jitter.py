#!/usr/bin/env python3
"""How the host-side stages of a resident encode vary from step to step: N encodes of one frame, the stage durations of
jxlt_last_frame_timeline per step -> median / 95th percentile / maximum per stage and the steps above 1.1 x the median.
Usage: stage_jitter.py [size] [steps]      (JXLT_POOL_MODE=0 / 1: never / always share the clustering, tools/code_probe.sh)"""
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
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    pkg = __graft_entry__.load_package()
    frame = bench.make_frame_on_device(torch, size, 0, torch.device("cuda", 0))
    torch.cuda.synchronize()
    enc = pkg.Encoder(0)
    enc.set_device_image([frame[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=frame)
    for _ in range(30):
        enc.encode_resident(1.0, copy=False)
    rows = []
    for _ in range(steps):
        t0 = time.perf_counter()
        enc.encode_resident(1.0, copy=False)
        total = (time.perf_counter() - t0) * 1e3
        tl = pkg.last_frame_timeline()
        rows.append((total, tl["dc_histogram_ms"], tl["ac_histogram_ms"] - tl["dc_histogram_ms"], tl["codes_ms"] - tl["ac_histogram_ms"],
                     tl["sizes_ms"] - tl["codes_ms"], tl["done_ms"] - tl["sizes_ms"]))
    names = ("step", "until DC histogram", "DC -> AC histogram", "codes", "sizes", "hand-over")
    for k, name in enumerate(names):
        v = sorted(r[k] for r in rows)
        print("%-20s median %.3f  p95 %.3f  max %.3f" % (name, v[len(v) // 2], v[int(len(v) * 0.95)], v[-1]))
    med = sorted(r[0] for r in rows)[len(rows) // 2]
    slow = [(i, r) for i, r in enumerate(rows) if r[0] > 1.1 * med]
    print("%d of %d steps above 1.1 x the median; mean %.3f" % (len(slow), len(rows), sum(r[0] for r in rows) / len(rows)))
    for i, r in slow[:12]:
        print("  step %3d: %s" % (i, "  ".join("%s %.3f" % (n.split()[0], x) for n, x in zip(names, r))))


if __name__ == "__main__":
    main()
