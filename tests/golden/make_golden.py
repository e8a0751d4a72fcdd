#!/usr/bin/env python3
"""Generates the committed regression fixtures under tests/golden/.

PROVENANCE: these vectors are outputs of THIS repository's oracle
(oracle/jxl_tiny_oracle.c) + host back-end, not of the reference: the reference
ships no vectors for this path and cannot be built in this image (SURVEY.md F4,
F5), so they pin the oracle/kernels/back-end against regressions, nothing more.
Each fixture: a small seeded input (stored, float32) and the expected per-group
token streams, side-band grids and codestream bytes.
"""
import hashlib
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
import jxlt_testlib as T  # noqa: E402

FIXTURES = [
    # name, w, h, distance, hard, force_dct8
    ("smooth_96x72_d1", 96, 72, 1.0, False, False),
    ("smooth_200x137_d2", 200, 137, 2.0, False, False),
    ("noise_72x40_d0p5", 72, 40, 0.5, True, False),
    ("smooth_264x260_d8", 264, 260, 8.0, False, False),
    ("smooth_128x64_dct8", 128, 64, 1.0, False, True),
]


def main():
    for name, w, h, d, hard, dct8 in FIXTURES:
        img = T.synthetic_image(w, h, hard=hard)
        planes = T.to_planes(img)
        r = T.oracle_hot_path(planes, d, dct8)
        jxl = T.assemble_codestream(r, d) if not dct8 else b""
        np.savez_compressed(
            HERE / (name + ".npz"), planes=planes, distance=np.float32(d), force_dct8=np.bool_(dct8),
            quant_dc=r.quant_dc, raw_quant=r.raw_quant, strategy=r.strategy, ytox=r.ytox, ytob=r.ytob,
            tokens=np.frombuffer(r.all_tokens(), np.uint8),
            group_token_bytes=np.array([len(t) for t in r.group_tokens], np.int64),
            xyb=r.xyb, quant_field=r.qf, masking=r.mask,
            codestream=np.frombuffer(jxl, np.uint8))
        print(name, "tokens", len(r.all_tokens()) // 3, "jxl", len(jxl), hashlib.sha256(jxl).hexdigest()[:12])


if __name__ == "__main__":
    main()
