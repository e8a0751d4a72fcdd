#!/usr/bin/env python3
"""Longer version of tests/test_gpu_parity.py::test_random_sequences_of_calls_on_one_context: random sequences of
C-ABI calls on one context, more frame shapes (incl. a single-group frame and one with three rows of DC groups),
three distances, frames from device memory / page-locked planes / a page-locked PFM payload.
Usage: api_fuzz.py [seeds] [steps]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__  # noqa: E402
import jxlt_testlib as T  # noqa: E402

built = __graft_entry__.load_package()
T.build_oracle()
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
SHAPES = [(300, 264), (96, 72), (520, 2100), (24, 2048 + 2048 + 72), (1096, 840)]
DIST = [0.5, 1.0, 8.0]
frames = [T.to_planes(T.synthetic_image(w, h, seed=70 + i, hard=(i in (1, 4)))) for i, (w, h) in enumerate(SHAPES)]
want, toks = {}, {}
for i, p in enumerate(frames):
    for d in DIST:
        r = T.oracle_hot_path(p, d)
        want[i, d] = T.assemble_codestream(r, d)
        toks[i, d] = r.all_tokens()
pinned = []
for p in frames:
    a, owner = built.pinned_empty(p.shape)
    a[...] = p
    pay, owner2 = built.pinned_empty((p.size,))
    pay[...] = T.pfm_payload(p)
    pinned.append((a, owner, pay, owner2))
bad = 0
for seed in range(seeds):
    rng = np.random.default_rng(1000 + seed)
    enc = built.Encoder(0)
    enc.set_wait_mode(seed % 3 == 2)  # (every third seed: the context as a lane of a batch holds it)
    cur = 0
    enc.upload(frames[cur])
    encoded = None
    trail = []
    try:
        for step in range(steps):
            op = int(rng.integers(0, 10))
            d = DIST[int(rng.integers(0, 3))]
            trail.append((op, cur, d))
            if op == 0:
                cur = int(rng.integers(0, len(frames)))
                how = int(rng.integers(0, 3))
                if how == 0:
                    enc.upload(frames[cur])
                elif how == 1:
                    enc.attach_host(pinned[cur][0])
                else:
                    enc.attach_host_pfm(pinned[cur][2], SHAPES[cur][0], SHAPES[cur][1])
                encoded = None
            elif op in (1, 8):
                assert enc.encode_resident(d) == want[cur, d]
                encoded = d
            elif op in (2, 9):
                assert enc.encode_resident(d, copy=False).tobytes() == want[cur, d]
                encoded = d
            elif op == 3:
                assert enc.encode_resident_raw_tokens(d) == want[cur, d]
                encoded = d
            elif op == 4:
                enc.enqueue(d, 0)
                assert built.HotPathOutput(enc.fetch_raw()).all_tokens() == toks[cur, d]
                encoded = d
            elif op == 5 and encoded is not None:
                ac, dc = enc.fetch_histograms()
                at, dt = built.build_code_tables(ac, dc)
                kind = int(rng.integers(0, 2))
                data, off, bits = enc.pack_sections(kind, at if kind else dt)
                assert len(data) == int(off[-1]) and ((bits + 7) // 8 == np.diff(off)).all()
                if kind == 1 and cur != 1:
                    assert want[cur, encoded].endswith(data.tobytes())
            elif op == 6 and encoded is not None:
                enc.stats()
            elif op == 7 and encoded is not None and seed % 3 != 2:  # (a batch lane's frames carry no stage events)
                enc.kernel_times()
        print("seed %d: %d steps ok" % (seed, steps), flush=True)
    except AssertionError:
        bad += 1
        print("seed %d: MISMATCH at step %d, last calls %s" % (seed, len(trail) - 1, trail[-6:]), flush=True)
    enc.close()
print("mismatching seeds:", bad)
sys.exit(1 if bad else 0)
