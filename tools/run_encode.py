#!/usr/bin/env python3
"""Minimal driver for profilers: N device-only passes of the hot path on a synthetic frame.
Usage: run_encode.py [size] [passes]"""
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
    pkg = __graft_entry__.load_package()
    dev = torch.device("cuda", 0)
    frame = bench.frame_rows_on_device(torch, size, 0, size, 0, dev)
    torch.cuda.synchronize()
    enc = pkg.Encoder(0)
    enc.set_device_image([frame[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=frame)
    for _ in range(passes):
        enc.enqueue(1.0, 0)
        enc.synchronize()
    print("done", size, passes)


if __name__ == "__main__":
    main()
