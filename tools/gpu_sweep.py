#!/usr/bin/env python3
"""One-off extended parity sweep on the GPU box: N random geometries / distances / seeds (incl. sizes
up to 2600 px so that several DC groups occur), codestream of the drop-in vs oracle + host assembler,
every codestream also read back by the independent decoder.  Usage: gpu_sweep.py [N] [seed]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__  # noqa: E402
import jxl_decoder as D  # noqa: E402
import jxlt_testlib as T  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    pkg = __graft_entry__.load_package()
    bad = 0
    t0 = time.time()
    for i in range(n):
        big = i % 8 == 0
        w = int(rng.integers(2049, 2600)) if big else int(rng.integers(9, 1100))
        h = int(rng.integers(9, 400)) if big else int(rng.integers(9, 900))
        d = float(np.round(10 ** rng.uniform(-1.3, 1.3), 3))
        seed = int(rng.integers(0, 1 << 30))
        planes = T.to_planes(T.synthetic_image(w, h, seed=seed, hard=(seed % 3 == 0)))
        want = T.assemble_codestream(T.oracle_hot_path(planes, d), d)
        got = pkg.encode_file(planes, d)
        ok = got == want
        psnr = float("nan")
        if ok and w * h < 500000:
            psnr = D.psnr_opsin_db(planes, D.decode(got).linear_rgb)
        bad += not ok
        print("%4d x %4d d=%-7g seed=%-10d %s  %6d bytes  psnr %.1f" % (w, h, d, seed, "ok" if ok else "MISMATCH", len(got), psnr),
              flush=True)
    print("sweep: %d cases, %d mismatching, %.0f s" % (n, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
