#!/bin/bash
# VALU-utilisation counters of tile_kernel (two --pmc passes, counters only).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_util
for set in "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" ; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_util/$tag -- python3 tools/run_encode.py ${1:-8192} 2 > gpurun_out/pmc_util_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_util/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ("tile_kernel" in r["Kernel_Name"] or "tile12_kernel" in r["Kernel_Name"]) and "redo" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print("%-28s %16.0f" % (k, sum(agg[k]) / len(agg[k])))
PY
