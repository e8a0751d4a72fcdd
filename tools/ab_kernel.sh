#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
IFS=';' read -ra VARIANTS <<< "$1"
for v in "${VARIANTS[@]}"; do
  [ "$v" = "base" ] && v=""
  touch libjxl-tiny_amd/csrc/jxlt_device_common.h
  make -C libjxl-tiny_amd -s -j3 csrc/libjxltiny_hip.so HIPFLAGS_EXTRA="$v" 2>&1 | grep -i error
  echo "== [$v]"
  for rep in 1 2; do ./tools/pack_cycles.sh ${SIZE:-16384} ${FILTER:-token_kernel}; done
done
touch libjxl-tiny_amd/csrc/jxlt_device_common.h
make -C libjxl-tiny_amd -s -j3 csrc/libjxltiny_hip.so 2>&1 | grep -i error
