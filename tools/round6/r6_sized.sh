#!/bin/bash
# Round 6: the sized form of the AC sections -- parity, then A/B against the two-pass form at the sizes that use it.
mkdir -p gpurun_out/r06m
timeout 1500 python -m pytest tests -m gpu -x -q -k "every_form or config4 or config3 or hot_path or random_frames or api_fuzz or call_sequence or sharded or multi" 2>&1 | tail -3
for rep in 1 2; do
for cfg in "JXLT_PACK_SIZED=0" "JXLT_PACK_SIZED=1"; do
  for sz in 16384 8192 4096; do echo -n "[$cfg] "; env $cfg timeout 300 python tools/run_resident.py $sz 40 2>&1 | grep done | cut -c1-120; done
done
done 2>&1 | tee gpurun_out/r06m/sized_ab.txt
echo "--- d=0.5 and noise"
for cfg in "JXLT_PACK_SIZED=0" "JXLT_PACK_SIZED=1"; do
  echo -n "[$cfg] "; env $cfg timeout 300 python tools/run_resident.py 16384 16 0.5 2>&1 | grep done | cut -c1-120
  echo -n "[$cfg] "; env $cfg timeout 300 python tools/run_resident.py 8192 16 1.0 noise 2>&1 | grep done | cut -c1-120
done 2>&1 | tee -a gpurun_out/r06m/sized_ab.txt
