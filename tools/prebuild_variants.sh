#!/bin/bash
# Builds libjxltiny_hip.so variants HERE (the build container) into gpurun_tmp/variants/<n>.so, so that a GPU call only
# swaps files (a rebuild on the GPU box costs a minute of box time per variant).  Usage: prebuild_variants.sh "base;-DX=1;-DY"
# The production build is restored at the end.
cd "$(dirname "$0")/.."
mkdir -p gpurun_tmp/variants
rm -f gpurun_tmp/variants/*.so gpurun_tmp/variants/list.txt
IFS=';' read -ra VARIANTS <<< "$1"
i=0
for v in "${VARIANTS[@]}"; do
  [ "$v" = "base" ] && v=""
  touch libjxl-tiny_amd/csrc/jxlt_device_common.h
  make -C libjxl-tiny_amd -s -j8 csrc/libjxltiny_hip.so HIPFLAGS_EXTRA="$v" 2>&1 | grep -iE "error" 
  cp libjxl-tiny_amd/csrc/libjxltiny_hip.so gpurun_tmp/variants/$i.so
  echo "$i [$v]" >> gpurun_tmp/variants/list.txt
  i=$((i+1))
done
touch libjxl-tiny_amd/csrc/jxlt_device_common.h
make -C libjxl-tiny_amd -s -j8 csrc/libjxltiny_hip.so 2>&1 | grep -iE "error"
cat gpurun_tmp/variants/list.txt
