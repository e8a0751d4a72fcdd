#!/bin/bash
# Step time of the bench frame against the hand-over's knobs: workgroups per hand-over kernel, writing launches and
# the growth of their shares, the DC-group sections' hand-over at once or behind the AC measuring pass.
cd "${GRAFT_REPO_ROOT:-.}"
SIZE=${1:-16384}
run() { echo "== $*"; env "$@" python3 tools/run_resident.py $SIZE 24 1.0 2>&1 | tail -1; }
for wgs in 4 8 16 32; do run JXLT_DELIVER_WGS=$wgs; done
for wgs in 8 16; do run JXLT_DELIVER_WGS=$wgs JXLT_DC_DELIVER_AFTER_AC_MEASURE=1; done
for cfg in "4 200" "4 150" "4 100" "6 130" "6 100" "8 100"; do
  set -- $cfg
  run JXLT_DELIVER_WGS=8 JXLT_PACK_LAUNCHES=$1 JXLT_PACK_GROWTH=$2
done
run JXLT_DELIVER_WGS=8 JXLT_PACK_LAUNCHES=4 JXLT_PACK_GROWTH=100 JXLT_DC_DELIVER_AFTER_AC_MEASURE=1
