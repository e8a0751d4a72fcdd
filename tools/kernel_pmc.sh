#!/bin/bash
# Per-kernel counters of complete resident encodes.  Usage: kernel_pmc.sh <size> <name filter> <counter sets...>
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
SIZE=$1; FILTER=$2; shift 2
rm -rf gpurun_out/kpmc
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/kpmc/$i -- python3 tools/run_resident.py $SIZE 2 > gpurun_out/kpmc_$i.log 2>&1
done
python3 - "$FILTER" <<'PY'
import csv, glob, collections, sys
flt = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/kpmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-30:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in agg:
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("   %-28s sum %16.0f  calls %d" % (c, sum(v) / 2, len(v) // 2))   # per encode (2 passes)
PY
