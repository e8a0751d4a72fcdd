#!/bin/bash
mkdir -p gpurun_out/r06p
timeout 900 python -m pytest tests -m gpu -x -q -k "every_form or config4" 2>&1 | tail -2
for rep in 1 2; do
  for cfg in "JXLT_PACK_SIZED=0" "JXLT_PACK_SIZED=1"; do
    echo "== $cfg"
    env $cfg bash tools/pack_cycles.sh 16384 jxlt_dev 2>&1 | grep -E "pack_tile_sized_kernel|pack_tile_write|pack_tile_measure|section_sizes" | cut -c1-100
    for sz in 16384 8192 4096; do echo -n "   "; env $cfg timeout 300 python tools/run_resident.py $sz 40 2>&1 | grep done | cut -c1-60; done
  done
done 2>&1 | tee gpurun_out/r06p/sized_v3_ab.txt
for cfg in "JXLT_PACK_SIZED=0" "JXLT_PACK_SIZED=1"; do
  echo -n "[$cfg d=0.5] "; env $cfg timeout 300 python tools/run_resident.py 16384 16 0.5 2>&1 | grep done | cut -c1-60
  echo -n "[$cfg noise] "; env $cfg timeout 300 python tools/run_resident.py 8192 16 1.0 noise 2>&1 | grep done | cut -c1-60
done 2>&1 | tee -a gpurun_out/r06p/sized_v3_ab.txt
