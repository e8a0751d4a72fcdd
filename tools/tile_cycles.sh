#!/bin/bash
# Shader-clock cycles and duration of tile_kernel (GRBM_GUI_ACTIVE / 8 XCDs), clock-independent A/B metric.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
rm -rf gpurun_out/clk
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/clk -- python3 tools/run_encode.py ${1:-16384} 6 > gpurun_out/clk.log 2>&1
python3 - <<PY
import csv,glob
k={}
for f in glob.glob("gpurun_out/clk/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k[r["Dispatch_Id"]]=(r["Kernel_Name"], int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
c={}
for f in glob.glob("gpurun_out/clk/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        c.setdefault(r["Dispatch_Id"],{})[r["Counter_Name"]]=float(r["Counter_Value"])
import collections
agg=collections.defaultdict(list)
for d,(name,ns) in k.items():
    if d in c: agg[name.split("(")[0]].append((ns, c[d]["GRBM_GUI_ACTIVE"]/8, c[d]["SQ_INSTS_VALU"]/max(1,c[d]["SQ_WAVES"])))
for name,v in agg.items():
    v=v[len(v)//2:]
    print("%-40s %8.3f ms %10.3f Mcycles  %7.0f VALU/wave" % (name[-40:], sum(x[0] for x in v)/len(v)/1e6, sum(x[1] for x in v)/len(v)/1e6, sum(x[2] for x in v)/len(v)))
    if name.endswith("tile12_kernel"):
        # the profile bench.py's roofline.valu_issue reads (copy it to profiles/<round>_tile_valu_<size>.json)
        import json, sys
        sys.path.insert(0, ".")
        import bench
        size = int("${1:-16384}")
        doc = {"frame": [size, size], "kernel": "tile12_kernel",
               "valu_insts_per_wave": round(sum(x[2] for x in v)/len(v)), "waves": (size // 64) * (size // 64) * 12,
               "shader_mcycles": round(sum(x[1] for x in v)/len(v)/1e6, 3), "kernel_ms": round(sum(x[0] for x in v)/len(v)/1e6, 3),
               "source": "tools/tile_cycles.sh %d (rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES --kernel-trace, counters-only run)" % size,
               "kernel_source_sha16": bench.kernel_source_sha16()}
        json.dump(doc, open("gpurun_out/clk/tile_valu_%d.json" % size, "w"), indent=1)
PY
