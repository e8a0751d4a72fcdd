#!/bin/bash
# Round 6: kernels of the packing stage in the two-pass and the sized form (cycles per encode), and the host's stage times.
mkdir -p gpurun_out/r06n
for cfg in "JXLT_PACK_SIZED=0" "JXLT_PACK_SIZED=1"; do
  echo "== $cfg"
  env $cfg bash tools/pack_cycles.sh 16384 jxlt_dev 2>&1 | grep -E "pack_|token_kernel|group_scan|publish"
  env $cfg bash tools/pack_cycles.sh 16384 fillBuffer 2>&1 | tail -2
done 2>&1 | tee gpurun_out/r06n/pack_cycles_ab.txt
for cfg in "JXLT_PACK_SIZED=0" "JXLT_PACK_SIZED=1"; do
  echo "== $cfg"
  env $cfg JXLT_TRACE_EVENTS=1 timeout 300 python tools/run_resident.py 16384 6 2>&1 | grep -E "jxlt event" | tail -40
done 2>&1 | tee gpurun_out/r06n/events_ab.txt | tail -90
