#!/bin/bash
# Kernel + copy timeline of one resident encode of a small frame (default 4096^2): where the latency goes.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
SIZE=${1:-4096}
rm -rf gpurun_out/tls
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tls -- python3 tools/run_resident.py $SIZE 12 > gpurun_out/tls.log 2>&1
python3 - <<'PY'
import csv,glob
ev=[]
for f in glob.glob("gpurun_out/tls/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][-30:]))
for f in glob.glob("gpurun_out/tls/**/*memory_copy_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY %s" % r.get("Direction","")))
ev.sort()
is_tile=lambda n: "tile12_kernel" in n and "redo" not in n
idx=[i for i,e in enumerate(ev) if is_tile(e[2])]
i0=idx[-2]
t0=ev[i0][0]
for k,(s,e,n) in enumerate(ev[i0-2:idx[-1]]):
    print("%9.3f %9.3f  %s" % ((s-t0)/1e6,(e-s)/1e6,n))
PY
