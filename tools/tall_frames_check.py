#!/usr/bin/env python3
"""Parity of narrow / tall and wide / flat frames at the edge of the single pass's range (up to 1024 groups in one
column or row; many DC-group sections) against the oracle, with the default packing form and with the other one forced."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import hashlib
    import numpy as np
    import __graft_entry__ as G
    import jxlt_testlib as T
    pkg = G.load_package()
    for (w, h, d) in [(256, 262144, 1.0), (262144, 200, 2.0), (300, 150000, 0.5), (8, 40000, 1.0), (50000, 9, 1.0)]:
        img = T.to_planes(T.synthetic_image(w, h, seed=w ^ h))
        e = pkg.Encoder(0)
        e.upload(img)
        a = bytes(e.encode_resident(d))
        b = bytes(e.encode_resident(d))
        e.close()
        print("RESULT %d %d %s %s" % (w, h, hashlib.sha256(a).hexdigest()[:16], "repeat-ok" if a == b else "REPEAT-DIFFERS"), flush=True)
        if os.environ.get("WITH_ORACLE") == "1":
            ref = T.oracle_encode_file(img, d, nthreads=16)[0]
            print("ORACLE %d %d %s" % (w, h, hashlib.sha256(bytes(ref)).hexdigest()[:16]), flush=True)
    sys.exit(0)

outs = {}
for name, env in [("default", {"WITH_ORACLE": "1"}), ("two passes", {"JXLT_PACK_TWO_PASS": "1"}), ("one pass", {"JXLT_PACK_TWO_PASS": "0"})]:
    r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=3000)
    outs[name] = [l for l in r.stdout.splitlines() if l.startswith(("RESULT", "ORACLE"))]
    print(name, "rc", r.returncode)
    for l in outs[name]:
        print("  ", l)
    if r.returncode != 0:
        print(r.stderr[-2000:])
want = {tuple(l.split()[1:3]): l.split()[3] for l in outs["default"] if l.startswith("ORACLE")}
bad = 0
for name, lines in outs.items():
    for l in lines:
        if l.startswith("RESULT"):
            f = l.split()
            if want.get((f[1], f[2])) != f[3] or f[4] != "repeat-ok":
                bad += 1
                print("MISMATCH", name, l)
print("mismatching:", bad)
sys.exit(1 if bad else 0)
