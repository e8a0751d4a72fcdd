#!/bin/bash
# Shader cycles (GRBM_GUI_ACTIVE / 8 XCDs), VALU / LDS instructions per wave of every kernel of complete resident
# encodes, summed per encode -- the A/B metric for the section-packing kernels.  Usage: pack_cycles.sh [size] [filter]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
rm -rf gpurun_out/pk
N=4
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d gpurun_out/pk -- python3 tools/run_resident.py ${1:-16384} $N > gpurun_out/pk.log 2>&1
python3 - "${2:-pack}" $N <<'PY'
import csv, glob, collections, sys
flt, n = sys.argv[1], int(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob("gpurun_out/pk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0][-28:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES": calls[k] += 1
for k, c in agg.items():
    w = max(1.0, c["SQ_WAVES"])
    print("%-28s %2d launches/encode %8.3f Mcycles/encode  %6.0f VALU/wave %5.0f LDS/wave  %6.0f bank-conflict cycles/wave  (%d launches seen, %.3f Mcycles each)" % (
        k, calls[k] // n, c["GRBM_GUI_ACTIVE"] / 8 / n / 1e6, c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_LDS"] / w, c["SQ_LDS_BANK_CONFLICT"] / w,
        calls[k], c["GRBM_GUI_ACTIVE"] / 8 / max(1, calls[k]) / 1e6))
PY
