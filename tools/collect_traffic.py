#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE CSVs of tools/run_resident.py (complete encodes; and of the
calibration binary tools/fetch_calib) into profiles/<tag>_traffic.json.

Method (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are collected in
separate --pmc passes; both count KiB at the L2's memory side.  On gfx950 FETCH_SIZE reads
half the bytes of a coalesced stream -- the calibration pass (1 GiB read with the same
4-byte-per-lane loads, 1 GiB written with the same 2-byte-per-lane stores) measures the
correction factors, which are then applied to the kernels' counters.

Usage: collect_traffic.py <calib_fetch.csv> <calib_write.csv> <fetch.csv> <write.csv> <size> <out.json> [encodes]"""
import collections
import csv
import json
import sys
from pathlib import Path


def per_kernel(path, per=1):
    """Counter value per kernel name: the average per launch (per = 1) or the sum over the launches of one
    encode (per = the number of encodes the run made)."""
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    if per == 1:
        return {k: sum(v) / len(v) for k, v in agg.items()}
    return {k: sum(v) / per for k, v in agg.items()}


def find(d, needle):
    for k, v in d.items():
        if needle in k:
            return v
    raise KeyError(needle)


def main():
    cf, cw, f, w, size, out = sys.argv[1:7]
    encodes = int(sys.argv[7]) if len(sys.argv) > 7 else 3  # complete encodes of the profiled run
    size = int(size)
    gib_kib = float(1 << 20)
    fetch_corr = gib_kib / find(per_kernel(cf), "read_dword_per_lane")
    write_corr = gib_kib / find(per_kernel(cw), "write_short_per_lane")
    res = {"frame": [size, size], "unit": "bytes per encode (all launches of the kernel in one complete encode)",
           "fetch_correction": round(fetch_corr, 4), "write_correction": round(write_corr, 4), "kernels": {}}
    fk, wk = per_kernel(f, encodes), per_kernel(w, encodes)
    names = ("tile12_kernel(", "tile_kernel(", "tile12_kernel_redo", "tile_kernel_redo", "token_kernel",
             "dc_elementwise_kernel", "dc_chain_summary_kernel", "dc_chain_kernel", "pack_tile_measure_kernel",
             "pack_tile_write_kernel", "pack_tile_count_kernel", "pack_tile_plan_kernel", "pack_tile_offsets_kernel",
             "pack_tile_finalize_kernel", "group_scan_kernel")
    for name in names:
        try:
            rd = find(fk, name) * 1024 * fetch_corr
            wr = find(wk, name) * 1024 * write_corr
        except KeyError:
            continue
        key = name.rstrip("(")
        res["kernels"][key] = {"hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr), "hbm_bytes": round(rd + wr)}
    # (bench.py's roofline.traffic reads "tile_kernel": the variant the product launches)
    if "tile12_kernel" in res["kernels"] and "tile_kernel" not in res["kernels"]:
        res["kernels"]["tile_kernel"] = dict(res["kernels"]["tile12_kernel"], variant="tile12_kernel (12 waves)")
    # (bench.py's roofline.traffic_profile compares this with the sources it runs)
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench
    res["kernel_source_sha16"] = bench.kernel_source_sha16()
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
