#!/usr/bin/env python3
"""Per-frame time of resident encodes of a small frame, a new context per round (tools/config_table.py's protocol):
how stable is it?  Usage: small_probe.py [size] [rounds]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__  # noqa: E402
import bench  # noqa: E402
import torch  # noqa: E402

pkg = __graft_entry__.load_package()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
f = bench.make_frame_on_device(torch, size, 0, torch.device("cuda", 0))
torch.cuda.synchronize()
for r in range(rounds):
    enc = pkg.Encoder(0)
    enc.set_device_image([f[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=f)
    for _ in range(6):
        enc.encode_resident(1.0, copy=False)
    ts = []
    for _ in range(16):
        t0 = time.perf_counter()
        enc.encode_resident(1.0, copy=False)
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print("round %d: mean %.3f ms, min %.3f, median %.3f, max %.3f" % (r, sum(ts) / len(ts), ts[0], ts[8], ts[-1]), flush=True)
    enc.close()
