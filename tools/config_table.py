#!/usr/bin/env python3
"""One-GPU timings of the workloads BASELINE.json lists (its configs #2-#5 shapes) plus the
token-heavy uniform-noise frame of SURVEY.md 8(d).  Frames resident in HBM unless stated.
Prints one JSON object per line; tools/gpu_check.sh-style usage:  python3 tools/config_table.py"""
import json
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

FORCE_DCT8 = 1


def noise_frame(torch, size, device):
    gen = torch.Generator(device=device)
    gen.manual_seed(4321)
    return torch.rand((3, size, size), dtype=torch.float32, device=device, generator=gen)


def main():
    import torch
    pkg = __graft_entry__.load_package()
    dev = torch.device("cuda", 0)
    rows = []

    def resident(name, frame, size, distance=1.0, reps=16):
        enc = pkg.Encoder(0)
        enc.set_device_image([frame[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=frame)
        out = None
        # (six untimed encodes: buffers reach their size, the host's clustering has tried both of its ways)
        for _ in range(6):
            out = enc.encode_resident(distance, copy=False)
        t0 = time.perf_counter()
        for _ in range(reps):
            out = enc.encode_resident(distance, copy=False)
        dt = (time.perf_counter() - t0) / reps
        kt = enc.kernel_times()
        st = enc.stats()
        fr = enc.fetch_raw()
        tok = int(fr.group_token_offset[fr.num_groups])
        rows.append({"workload": name, "ms_per_frame": round(dt * 1e3, 3), "mpix_s": round(size * size / dt / 1e6, 1),
                     "codestream_bytes": len(out), "token_bytes_per_pixel": round(tok / (size * size), 3),
                     "tiles_redone_exact_roots": "%d of %d" % (st["tiles_redone_exact_roots"], st["tiles"]),
                     "kernel_ms": {k: round(v, 3) for k, v in kt.items()}})
        enc.close()

    def device_only(name, frame, size, flags, reps=16):
        enc = pkg.Encoder(0)
        enc.set_device_image([frame[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=frame)
        for _ in range(2):
            enc.enqueue(1.0, flags)
            enc.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            enc.enqueue(1.0, flags)
        enc.synchronize()
        dt = (time.perf_counter() - t0) / reps
        kt = enc.kernel_times()
        rows.append({"workload": name, "ms_per_frame": round(dt * 1e3, 3), "mpix_s": round(size * size / dt / 1e6, 1),
                     "kernel_ms": {k: round(v, 3) for k, v in kt.items()}})
        enc.close()

    f = bench.make_frame_on_device(torch, 4096, 0, dev)
    device_only("config #2: 4096x4096, fixed DCT8 strategy, device pipeline only (tokens + histograms in HBM)", f, 4096, FORCE_DCT8)
    resident("4096x4096 full search, .jxl bytes in host memory", f, 4096)
    del f
    f = bench.make_frame_on_device(torch, 8192, 0, dev)
    resident("config #3: 8192x8192 full search + adaptive quantisation, .jxl bytes in host memory", f, 8192)
    del f
    f = bench.make_frame_on_device(torch, 16384, 0, dev)
    resident("config #4 (one GPU's view): 16384x16384, .jxl bytes in host memory", f, 16384)
    resident("16384x16384 at distance 0.5", f, 16384, distance=0.5, reps=10)
    resident("16384x16384 at distance 4", f, 16384, distance=4.0, reps=10)
    resident("16384x16384 at distance 0.1", f, 16384, distance=0.1, reps=6)
    f *= 4.0  # samples up to 4.0 (the reference documents values outside [0, 1] as legal)
    resident("16384x16384 HDR (the frame's samples x 4: up to 4.0), distance 1", f, 16384, reps=10)
    resident("16384x16384 HDR (samples up to 4.0) at distance 0.1", f, 16384, distance=0.1, reps=6)
    del f
    f = noise_frame(torch, 8192, dev)
    resident("token-heavy: 8192x8192 uniform noise (SURVEY 8(d) 'hard' set)", f, 8192, reps=10)
    del f
    for r in rows:
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
