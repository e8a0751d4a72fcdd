#!/bin/bash
mkdir -p gpurun_out/r06q
for rep in 1 2; do
for cfg in "JXLT_PACK_SIZED=0" "JXLT_PACK_SIZED=1"; do
  echo -n "[$cfg] "; env $cfg timeout 600 python tools/slab_of_8.py 16384 30 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_frame_median'], d['ms_per_frame_min'], d['stage_split_ms_median'], d['same_bytes_as_single_gpu'])"
done
done 2>&1 | tee gpurun_out/r06q/slab8_sized_ab.txt
