#!/bin/bash
# Which command of an early frame takes 7 ms?  Kernel + memory-copy trace of the first eight encodes of a context;
# every command longer than 1 ms that is not tile12_kernel, and every memory copy of more than 8 MB.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
rm -rf gpurun_out/sc
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/sc -- python3 tools/run_resident.py 16384 8 > gpurun_out/sc.log 2>&1
python3 - <<'PY'
import csv,glob
ev=[]
for f in glob.glob("gpurun_out/sc/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][-32:], r.get("Queue_Id","")))
ncopy=0
for f in glob.glob("gpurun_out/sc/**/*memory_copy_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        ncopy+=1
        ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY %s %s bytes" % (r.get("Direction",""), r.get("Bytes", r.get("Size",""))), ""))
ev.sort()
tiles=[e for e in ev if "tile12_kernel" in e[2] and "redo" not in e[2]]
t0=tiles[0][0]
print("memory-copy records:", ncopy, " tile kernels:", len(tiles))
print("tile kernel starts (ms):", [round((t[0]-t0)/1e6,2) for t in tiles])
for s,e,n,q in ev:
    dur=(e-s)/1e6
    big = n.startswith("COPY") and any(int(x) > (8<<20) for x in n.split() if x.isdigit())
    if (dur > 1.0 and "tile12_kernel" not in n) or big or ("copyBuffer" in n and dur > 0.15):
        print("%10.3f ms  %8.3f ms  %s %s" % ((s-t0)/1e6, dur, n, q))
PY
