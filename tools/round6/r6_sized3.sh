#!/bin/bash
mkdir -p gpurun_out/r06o
cp libjxl-tiny_amd/csrc/libjxltiny_hip.so gpurun_tmp/variants/production.so
for rep in 1 2; do
for v in 0 1; do
  cp gpurun_tmp/variants/$v.so libjxl-tiny_amd/csrc/libjxltiny_hip.so
  for cfg in "JXLT_PACK_SIZED=0" "JXLT_PACK_SIZED=1"; do
    echo "== variant $v $cfg"
    env $cfg bash tools/pack_cycles.sh 16384 jxlt_dev 2>&1 | grep -E "pack_tile_sized_kernel|pack_tile_write|pack_tile_measure|section_sizes" | cut -c1-100
    for sz in 16384 8192; do echo -n "   "; env $cfg timeout 300 python tools/run_resident.py $sz 40 2>&1 | grep done | cut -c1-60; done
  done
done
done 2>&1 | tee gpurun_out/r06o/sized_tiles_ab.txt
cp gpurun_tmp/variants/production.so libjxl-tiny_amd/csrc/libjxltiny_hip.so
