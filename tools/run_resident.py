#!/usr/bin/env python3
"""Minimal driver for profilers: N complete resident encodes (device pipeline + code
construction + packing + placement).  Usage: run_resident.py [size] [passes] [distance] [noise]
Prints the time per encode of the second half of the passes and the stage times of the last one."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__  # noqa: E402
import bench  # noqa: E402


def main():
    import torch
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    distance = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    pkg = __graft_entry__.load_package()
    if len(sys.argv) > 4 and sys.argv[4] == "noise":
        gen = torch.Generator(device="cuda:0")
        gen.manual_seed(4321)
        frame = torch.rand((3, size, size), dtype=torch.float32, device="cuda:0", generator=gen)
    else:
        frame = bench.make_frame_on_device(torch, size, 0, torch.device("cuda", 0))
    torch.cuda.synchronize()
    enc = pkg.Encoder(0)
    enc.set_device_image([frame[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=frame)
    import time
    n = 0
    t0 = None
    each = []
    for i in range(passes):
        if i == passes // 2:
            t0 = time.perf_counter()
        t1 = time.perf_counter()
        n = enc.encode_resident(distance, copy=False)
        each.append((time.perf_counter() - t1) * 1e3)
    ms = (time.perf_counter() - t0) / (passes - passes // 2) * 1e3
    half = sorted(each[passes // 2:])
    print("done", size, passes, len(n), "%.3f ms per encode" % ms, {k: round(v, 3) for k, v in enc.kernel_times().items()},
          "median %.3f min %.3f" % (half[len(half) // 2], half[0]))


if __name__ == "__main__":
    main()
