#!/usr/bin/env python3
"""Minimal driver for profilers: N complete resident encodes (device pipeline + code
construction + packing + placement).  Usage: run_resident.py [size] [passes]"""
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
    frame = bench.make_frame_on_device(torch, size, 0, torch.device("cuda", 0))
    torch.cuda.synchronize()
    enc = pkg.Encoder(0)
    enc.set_device_image([frame[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=frame)
    n = 0
    for _ in range(passes):
        n = enc.encode_resident(1.0, copy=False)
    print("done", size, passes, n)


if __name__ == "__main__":
    main()
