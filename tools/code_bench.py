#!/usr/bin/env python3
"""Host time of the code construction (histogram clustering + Huffman codes, host/entropy_coder.cc) on the histograms
of the 16384^2 bench frame at four distances (tests/golden/histograms/bench_histograms.npz, written by tools/dump_histograms.py on
the GPU box).  No GPU needed.  JXLT_POOL_MODE=0/1 forces the clustering to work alone / with the helper threads."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
H = np.load(ROOT / "tests" / "golden" / "histograms" / "bench_histograms.npz")
zero = np.zeros((64, 64), np.uint32)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for key in sorted(H.files):
    h = np.ascontiguousarray(H[key])
    args = (h, zero) if key.startswith("ac") else (zero, h)
    best, tot = 1e9, 0.0
    for _ in range(reps):
        t = time.perf_counter()
        tables = pkg.build_code_tables(*args)
        dt = time.perf_counter() - t
        best, tot = min(best, dt), tot + dt
    import hashlib
    print("%-8s %4d non-zero counts  best %.3f ms  mean %.3f ms  tables %s" % (
        key, int((h != 0).sum()), best * 1e3, tot / reps * 1e3,
        hashlib.sha256(tables[0].tobytes() + tables[1].tobytes()).hexdigest()[:12]))
