#!/bin/bash
# A/B of tile12_kernel build variants on the GPU box, cycles only: for every flag set in argument 1 (separated by
# ';', "base" = the production build) rebuild the HIP library and print tile_cycles' line (shader cycles per 16384^2
# launch, VALU per wave).  PARITY=1: the quick GPU parity tests for the FIRST variant.  The production build is
# restored at the end.
cd "${GRAFT_REPO_ROOT:-.}"
IFS=';' read -ra VARIANTS <<< "$1"
first=1
for v in "${VARIANTS[@]}"; do
  [ "$v" = "base" ] && v=""
  touch libjxl-tiny_amd/csrc/jxlt_device_common.h
  make -C libjxl-tiny_amd -s -j3 csrc/libjxltiny_hip.so HIPFLAGS_EXTRA="$v" 2>&1 | grep -i error
  echo "== [$v]"
  if [ -n "$PARITY" ] && [ -n "$first" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "hot_path or golden or random_frames or values_outside or redo or redone or config3" 2>&1 | tail -1; fi
  first=
  for rep in 1 ${REPS:+2}; do ./tools/tile_cycles.sh ${SIZE:-16384} | grep -E "tile12_kernel  "; done
done
touch libjxl-tiny_amd/csrc/jxlt_device_common.h
make -C libjxl-tiny_amd -s -j3 csrc/libjxltiny_hip.so 2>&1 | grep -i error
