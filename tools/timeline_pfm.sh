#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
rm -rf gpurun_out/tlp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tlp -- python3 tools/trace_pfm.py 16384 2 > gpurun_out/tlp.log 2>&1
python3 - <<'PY'
import csv,glob,collections
k=collections.Counter(); kd=collections.Counter()
for f in glob.glob("gpurun_out/tlp/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0][-30:]
        k[n]+=1; kd[n]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for n,c in k.most_common(12): print("%-32s calls %5d total %.2f ms" % (n,c,kd[n]/1e6))
m=collections.Counter(); md=collections.Counter(); mb=collections.Counter()
for f in glob.glob("gpurun_out/tlp/**/*memory_copy_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        d=r.get("Direction","?"); m[d]+=1; md[d]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for d,c in m.items(): print("copy %-28s calls %5d total %.2f ms" % (d,c,md[d]/1e6))
PY
tail -2 gpurun_out/tlp.log
