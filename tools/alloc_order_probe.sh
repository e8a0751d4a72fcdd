#!/bin/bash
# tile_kernel with the device context made before / after the frame (bench.py, BENCH_CONTEXT_FIRST): time, shader cycles
# and the address-translation counters of the L1 (does the slow placement miss in the TLB?).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for cfg in "BENCH_CONTEXT_FIRST=1" "BENCH_NORMAL=1"; do
  for set in ${SETS:-"GRBM_GUI_ACTIVE TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS TCP_UTCL1_THRASHING_STALL TCP_UTCL1_SERIALIZATION_STALL GRBM_UTCL2_BUSY"}; do
    rm -rf gpurun_out/ao
    env $cfg timeout 300 rocprofv3 --pmc ${set//,/ } --kernel-trace --output-format csv -d gpurun_out/ao -- python3 bench.py --no-extras --steps 3 --warmup 2 > gpurun_out/ao.log 2>&1
    python3 - "$cfg" <<'PY'
import csv, glob, sys, collections
k = {}
for f in glob.glob("gpurun_out/ao/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/ao/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tile12_kernel(" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
ms = [v[1] / 1e6 for v in k.values() if "tile12_kernel(" in v[0]]
print(sys.argv[1], "tile12_kernel %.3f ms" % (sum(ms[-3:]) / max(1, len(ms[-3:]))), {n: round(sum(v[-3:]) / len(v[-3:])) for n, v in agg.items()})
PY
  done
done
