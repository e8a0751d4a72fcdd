#!/usr/bin/env python3
"""One-off evidence on the GPU box: encode the benchmark frame (bench.make_frame_on_device) of the given
size on the GPU, then read the codestream back with the independent reader (tests/jxl_decoder.py):
every TOC section consumed to its last byte, token count, PSNR against the input.
Usage: decode_bench_frame.py [size=16384]   (a 16384^2 decode takes a few minutes and ~20 GB of host memory)"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__  # noqa: E402
import bench  # noqa: E402
import jxl_decoder as D  # noqa: E402


def main():
    import torch
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    pkg = __graft_entry__.load_package()
    frame = bench.make_frame_on_device(torch, size, 0, torch.device("cuda", 0))
    enc = pkg.Encoder(0)
    enc.set_device_image([frame[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=frame)
    jxl = bytes(enc.encode_resident(1.0, copy=True))
    planes = frame.cpu().numpy()
    del frame
    t = time.time()
    dec = D.decode(jxl)
    psnr = D.psnr_opsin_db(planes, dec.linear_rgb)
    print("%dx%d bench frame encoded on the GPU: %d bytes, %d sections, %d tokens; independent reader: every section "
          "consumed, PSNR (cube-root LMS) %.2f dB, decode %.0f s" % (size, size, len(jxl), len(dec.section_sizes),
                                                                  dec.num_tokens, psnr, time.time() - t))


if __name__ == "__main__":
    main()
