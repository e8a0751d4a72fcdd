#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE CSVs of tools/run_encode.py (and of the
calibration binary tools/fetch_calib) into profiles/<tag>_traffic.json.

Method (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are collected in
separate --pmc passes; both count KiB at the L2's memory side.  On gfx950 FETCH_SIZE reads
half the bytes of a coalesced stream -- the calibration pass (1 GiB read with the same
4-byte-per-lane loads, 1 GiB written with the same 2-byte-per-lane stores) measures the
correction factors, which are then applied to the kernels' counters.

Usage: collect_traffic.py <calib_fetch.csv> <calib_write.csv> <fetch.csv> <write.csv> <size> <out.json>"""
import collections
import csv
import json
import sys


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def find(d, needle):
    for k, v in d.items():
        if needle in k:
            return v
    raise KeyError(needle)


def main():
    cf, cw, f, w, size, out = sys.argv[1:7]
    size = int(size)
    gib_kib = float(1 << 20)
    fetch_corr = gib_kib / find(per_kernel(cf), "read_dword_per_lane")
    write_corr = gib_kib / find(per_kernel(cw), "write_short_per_lane")
    res = {"frame": [size, size], "unit": "bytes per launch",
           "fetch_correction": round(fetch_corr, 4), "write_correction": round(write_corr, 4), "kernels": {}}
    fk, wk = per_kernel(f), per_kernel(w)
    for name in ("tile_kernel", "token_kernel", "pack_kernel", "dc_elementwise_kernel", "dc_chain_kernel"):
        try:
            rd = find(fk, name) * 1024 * fetch_corr
            wr = find(wk, name) * 1024 * write_corr
        except KeyError:
            continue
        res["kernels"][name] = {"hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr),
                                "hbm_bytes": round(rd + wr)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
