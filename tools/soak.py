#!/usr/bin/env python3
"""Soak: many encodes in a row (one context, then contexts created and destroyed), watching the time per frame,
the process's resident set and the device memory in use.  Usage: soak.py [size] [frames]"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import os  # noqa: E402
# (contexts in a row: keep the device blocks of destroyed contexts beyond the last one -- an opt-in since round 4)
os.environ.setdefault("JXLT_DEVICE_CACHE_MB", "32768")
import __graft_entry__  # noqa: E402
import bench  # noqa: E402
import torch  # noqa: E402

pkg = __graft_entry__.load_package()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2000


def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2**20


def used_mb():
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / 2**20


f = bench.make_frame_on_device(torch, size, 0, torch.device("cuda", 0))
torch.cuda.synchronize()
enc = pkg.Encoder(0)
enc.set_device_image([f[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=f)
ref = enc.encode_resident(1.0)
print("start: rss %.0f MB, device %.0f MB" % (rss_mb(), used_mb()), flush=True)
for block in range(5):
    t0 = time.perf_counter()
    for i in range(frames // 5):
        out = enc.encode_resident(1.0, copy=False)
    dt = (time.perf_counter() - t0) / (frames // 5)
    assert out.tobytes() == ref
    print("one context, block %d: %.3f ms per frame, rss %.0f MB, device %.0f MB" % (block, dt * 1e3, rss_mb(), used_mb()), flush=True)
enc.close()
for block in range(5):
    t0 = time.perf_counter()
    for i in range(20):
        e = pkg.Encoder(0)
        e.set_device_image([f[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=f)
        for _ in range(5):
            out = e.encode_resident(1.0, copy=False)
        assert out.tobytes() == ref
        e.close()
    dt = (time.perf_counter() - t0) / 20
    print("20 contexts x 5 frames, block %d: %.2f ms per context, rss %.0f MB, device %.0f MB" % (block, dt * 1e3, rss_mb(), used_mb()), flush=True)
print("released", pkg.release_cached_memory() >> 20, "MB; device %.0f MB" % used_mb())
