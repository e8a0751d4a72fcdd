#!/bin/bash
mkdir -p gpurun_out/r06t
REPS=2 bash tools/run_variants.sh 2>&1 | tee gpurun_out/r06t/ilp_ab.txt
# the whole step and the other kernels with the max-ilp build (variant 2)
cp libjxl-tiny_amd/csrc/libjxltiny_hip.so gpurun_tmp/variants/production.so
for v in 0 2; do
  cp gpurun_tmp/variants/$v.so libjxl-tiny_amd/csrc/libjxltiny_hip.so
  echo "== variant $v"; timeout 300 python tools/run_resident.py 16384 40 2>&1 | grep done | cut -c1-120
  bash tools/pack_cycles.sh 16384 jxlt_dev 2>&1 | grep -E "token_kernel|pack_tile_write|pack_tile_measure|dc_" | cut -c1-110
done 2>&1 | tee gpurun_out/r06t/ilp_other_kernels.txt
cp gpurun_tmp/variants/production.so libjxl-tiny_amd/csrc/libjxltiny_hip.so
