#!/bin/bash
mkdir -p gpurun_out/r06w
timeout 900 python -m pytest tests -m gpu -x -q -k "every_form or batch or config5 or hot_path or random_frames or api_fuzz or call_sequence or error_behaviour or single_pass" 2>&1 | tail -2
for rep in 1 2; do
for lanes in 1 2 4 6 8 10; do
  echo -n "lanes $lanes: "
  timeout 300 python3 bench.py --frame-batch 48 --frame-size 3840x2160 --frames-resident --lanes $lanes --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'MP/s', d['config']['frames_per_s'], 'frames/s', d['parity_gate'])"
done
done 2>&1 | tee gpurun_out/r06w/batch_lanes.txt
for sz in 8192 4096 2048 1024; do timeout 300 python tools/run_resident.py $sz 60 2>&1 | grep done | cut -c1-100; done | tee gpurun_out/r06w/resident_small.txt
