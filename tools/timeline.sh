#!/bin/bash
# Kernel + memory-copy timeline of the last bench step (no counters).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
rm -rf gpurun_out/tl
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --no-extras --steps 3 --warmup 1 "$@" > gpurun_out/tl.log 2>&1
python3 - <<'PY'
import csv,glob
ev=[]
for f in glob.glob("gpurun_out/tl/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][-28:]))
for f in glob.glob("gpurun_out/tl/**/*memory_copy_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY %s %s"%(r.get("Direction",""),r.get("Bytes", r.get("Size","")))))
ev.sort()
# the second-to-last timed step: from its tile_kernel to the next one
idx=[i for i,e in enumerate(ev) if "measure" in e[2]]
i0=idx[len(idx)//2]
is_tile=lambda n: ("tile_kernel" in n or "tile12_kernel" in n) and "redo" not in n
while not is_tile(ev[i0][2]): i0-=1
t0=ev[i0][0]
for k,(s,e,n) in enumerate(ev[i0:i0+80]):
    if k and is_tile(n): break
    print("%9.3f %9.3f  %s" % ((s-t0)/1e6,(e-s)/1e6,n))
PY
