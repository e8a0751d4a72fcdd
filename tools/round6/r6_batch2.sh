#!/bin/bash
mkdir -p gpurun_out/r06v
for rep in 1 2; do
for lanes in 4 6 8; do
  echo -n "lanes $lanes: "
  timeout 300 python3 bench.py --frame-batch 48 --frame-size 3840x2160 --frames-resident --lanes $lanes --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'MP/s', d['config']['frames_per_s'], 'frames/s', d['parity_gate'])"
done
done 2>&1 | tee gpurun_out/r06v/batch_lanes_fused_plan.txt
echo "--- two processes on the one GPU, three and four lanes each"
for lanes in 3 4; do
JXLT_BENCH_ONE_DEVICE=1 timeout 600 python3 bench.py --gpus 2 --frame-batch 96 --frame-size 3840x2160 --frames-resident --lanes $lanes --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('2 processes x', $lanes, 'lanes:', d['value'], 'MP/s', d['config']['frames_per_s'], 'frames/s')"
done 2>&1 | tee -a gpurun_out/r06v/batch_lanes_fused_plan.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "every_form or batch or config5 or hot_path or random_frames or api_fuzz or call_sequence" 2>&1 | tail -2
