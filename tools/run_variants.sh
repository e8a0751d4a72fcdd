#!/bin/bash
# On the GPU box: tile12_kernel's cycles for every prebuilt variant of gpurun_tmp/variants (tools/prebuild_variants.sh),
# REPS times round robin; the production library is put back at the end.
cd "${GRAFT_REPO_ROOT:-.}"
cp libjxl-tiny_amd/csrc/libjxltiny_hip.so gpurun_tmp/variants/production.so
for rep in $(seq 1 ${REPS:-2}); do
  while read -r idx flags; do
    cp gpurun_tmp/variants/$idx.so libjxl-tiny_amd/csrc/libjxltiny_hip.so
    echo -n "$flags  "; ./tools/tile_cycles.sh ${SIZE:-16384} | grep -E "tile12_kernel  "
  done < gpurun_tmp/variants/list.txt
done
cp gpurun_tmp/variants/production.so libjxl-tiny_amd/csrc/libjxltiny_hip.so
