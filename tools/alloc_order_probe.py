#!/usr/bin/env python3
"""How much does the ORDER of device allocations (and frees) in a process decide how fast tile_kernel runs?
  a  frame (torch), then context            -- what bench.py does
  b  context, then frame                    -- 4.8 instead of 4.05 ms when first tried
  c  as b, without the copy warm-up of the context's creation (its 36 MB are allocated and freed in front of everything)
  d  as a, with a 36 MB allocation made and freed in front of everything
  e  context, then the frame made by bench.py's generator (frame_rows_on_device: 1024-row chunks, many temporaries)
  f  the frame made by bench.py's generator, then the context (= bench.py)
Usage: alloc_order_probe.py <a|b|c|d>   (one variant per process)"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
mode = sys.argv[1]
if mode == "c":
    os.environ["JXLT_COPY_WARMUP"] = "0"
import __graft_entry__  # noqa: E402
import bench  # noqa: E402
import torch  # noqa: E402

pkg = __graft_entry__.load_package()
dev = torch.device("cuda", 0)
size = 16384
torch.zeros(1, device=dev)
enc = None
if mode == "d":
    t = torch.empty(36 << 20, dtype=torch.uint8, device=dev)
    del t
    torch.cuda.empty_cache()
if mode in ("b", "c", "e"):
    enc = pkg.Encoder(0)
if mode in ("e", "f"):
    frame = bench.frame_rows_on_device(torch, size, 0, size, 0, dev)
else:
    frame = bench.make_frame_on_device(torch, size, 0, dev)
torch.cuda.synchronize()
if enc is None:
    enc = pkg.Encoder(0)
enc.set_device_image([frame[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=frame)
for _ in range(12):
    enc.encode_resident(1.0, copy=False)
t0 = time.perf_counter()
for _ in range(20):
    enc.encode_resident(1.0, copy=False)
dt = (time.perf_counter() - t0) / 20
print(mode, "%.3f ms per encode" % (dt * 1e3), {k: round(v, 3) for k, v in enc.kernel_times().items()},
      "frame planes at", [hex(frame[c].data_ptr()) for c in range(3)],
      "torch reserved %d MB, allocated %d MB" % (torch.cuda.memory_reserved(0) >> 20, torch.cuda.memory_allocated(0) >> 20))
