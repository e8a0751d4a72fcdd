#!/usr/bin/env python3
"""Per-phase dynamic instruction counts of tile_kernel.

Run under rocprofv3 with instruction counters:
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv \
      -d gpurun_out/phase_pmc -- python3 tools/phase_pmc.py run [size]
The script launches the hot path ten times, truncating tile_kernel after phase 0..9
(flags bits 8-11, profiling only), so the i-th tile_kernel dispatch carries the cumulative
counts up to phase i.  Then:
  python3 tools/phase_pmc.py report gpurun_out/phase_pmc
prints per-phase instructions per wave."""
import csv
import glob
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

PHASES = ["P0 load+XYB", "P1 AQ energy", "P2-3 erosion", "P4 modulations", "P6a 2-block DCTs", "P5a DCT8",
          "P6b entropy", "P7 decision", "P8 quantise", "P9 scan store", "P5b CfL"]
ORDER = [0, 1, 2, 3, 4, 5, 10, 6, 7, 8, 9]  # execution order of the phase slots


def run(size):
    import torch
    import __graft_entry__
    import bench
    pkg = __graft_entry__.load_package()
    frame = bench.make_frame_on_device(torch, size, 0, torch.device("cuda", 0))
    torch.cuda.synchronize()
    enc = pkg.Encoder(0)
    enc.set_device_image([frame[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=frame)
    for i in ORDER:
        enc.enqueue(1.0, (i + 1) << 8)
        enc.synchronize()


def report(d):
    """Per-phase deltas of every counter found (per wave for SQ_INSTS_*, totals otherwise)."""
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    tile = [r for r in rows if ("tile_kernel" in r["Kernel_Name"] or "tile12_kernel" in r["Kernel_Name"]) and "redo" not in r["Kernel_Name"]]
    by_set = {}
    # several rocprofv3 passes (one directory each) may be reported together: dispatch order within a pass
    for r in tile:
        by_set.setdefault(r["Counter_Name"], {})[int(r["Dispatch_Id"])] = float(r["Counter_Value"])
    names = sorted(by_set)
    series = {}
    for n in names:
        vals = [by_set[n][k] for k in sorted(by_set[n])]
        assert len(vals) % len(PHASES) == 0, (n, len(vals))
        series[n] = vals[-len(PHASES):]
    waves = series.get("SQ_WAVES")
    print("%-24s" % "phase" + "".join("%22s" % n[-21:] for n in names))
    prev = {n: 0.0 for n in names}
    for j, name in enumerate([PHASES[i] for i in ORDER]):
        line = "%-24s" % name
        for n in names:
            v = series[n][j]
            if n.startswith("SQ_INSTS") and waves:
                v = v / waves[j]
            line += "%22.1f" % ((v - prev[n]) if n != "SQ_WAVES" else v)
            prev[n] = v
        print(line)
    print("%-24s" % "total" + "".join("%22.1f" % prev[n] for n in names))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 4096)
    else:
        report(sys.argv[2])
