#!/usr/bin/env python3
"""The 16384^2 resident encode in a process that never imports torch: the frame comes from numpy, is uploaded once
through the library (jxlt_image_upload) and encoded repeatedly.  Is tile12_kernel any faster (or slower) when the
library's streams and allocations are the only ones in the process?  Usage: no_torch_probe.py [size] [passes]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 40
pkg = __graft_entry__.load_package()
rng = np.random.default_rng(5)
yy, xx = np.mgrid[0:1024, 0:size].astype(np.float32)
rows = []
for y0 in range(0, size, 1024):
    base = np.stack([0.5 + 0.4 * np.sin(xx / 37) * np.cos((yy + y0) / 53), 0.5 + 0.4 * np.sin((xx + yy + y0) / 91),
                     0.3 + 0.3 * np.cos(xx / 19 - (yy + y0) / 29)]).astype(np.float32)
    base += rng.normal(0, 0.02, base.shape).astype(np.float32)
    rows.append(np.clip(base, 0, 1))
planes = np.ascontiguousarray(np.concatenate(rows, axis=1))
assert "torch" not in sys.modules
enc = pkg.Encoder(0)
enc.upload(planes)
n = 0
for i in range(passes):
    if i == passes // 2:
        t0 = time.perf_counter()
    n = enc.encode_resident(1.0, copy=False)
ms = (time.perf_counter() - t0) / (passes - passes // 2) * 1e3
print("no torch:", size, len(n), "%.3f ms per encode" % ms, {k: round(v, 3) for k, v in enc.kernel_times().items()}, "torch imported:", "torch" in sys.modules)
