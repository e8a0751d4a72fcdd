#!/usr/bin/env python3
"""How the oracle's pixel pipeline (C, POSIX threads over the groups) scales with threads on this host, next to what the
host says about itself (CPUs in the affinity mask, the cgroup's CPU quota, load).  No GPU needed."""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import jxlt_testlib as T  # noqa: E402

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "loadavg", os.getloadavg())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError:
        pass
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
img = T.to_planes(T.synthetic_image(size, size))
for n in (1, 8, 16, 32, 64, 128, 256):
    if n > 2 * (os.cpu_count() or 1):
        break
    t0 = time.perf_counter()
    out = T.oracle_encode_file(img, 1.0, nthreads=n)
    print("%3d threads: pixel pipeline %.3f s (%.1f Mpixel/s), bitstream %.3f s" % (n, out[1], size * size / out[1] / 1e6, out[2]), flush=True)
