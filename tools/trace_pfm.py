#!/usr/bin/env python3
"""Host-side timeline (JXLT_TRACE) of the PCIe-inclusive encode: a size x size PFM payload in page-locked host memory
-> jxlt_image_attach_host_pfm -> jxlt_encode_resident_view.  Usage: JXLT_TRACE=1 trace_pfm.py [size] [reps]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__  # noqa: E402
import bench  # noqa: E402


def main():
    import numpy as np
    import torch
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    pkg = __graft_entry__.load_package()
    dev = torch.device("cuda", 0)
    frame = bench.frame_rows_on_device(torch, size, 0, size, 0, dev)
    payload, owner = pkg.pinned_empty((size * size * 3,), np.float32)
    view = payload.reshape(size, size, 3)
    for r0 in range(0, size, 2048):
        blk = frame[:, r0:r0 + 2048].permute(1, 2, 0).flip(0).contiguous().cpu().numpy()
        view[size - r0 - blk.shape[0]:size - r0] = blk
    enc = pkg.Encoder(0)
    for i in range(reps):
        t0 = time.perf_counter()
        enc.attach_host_pfm(payload, size, size)
        out = enc.encode_resident(1.0, copy=False)
        t1 = time.perf_counter()
        print("rep %d: %.2f ms, %d bytes, kernel times %s" % (i, (t1 - t0) * 1e3, len(out), enc.kernel_times()), flush=True)


if __name__ == "__main__":
    main()
