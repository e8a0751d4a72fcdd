#!/bin/bash
mkdir -p gpurun_out/r06h
REPS=2 bash tools/run_variants.sh 2>&1 | tee gpurun_out/r06h/probes_p0.txt
for lanes in 1 2 4 6 8 12; do
  echo -n "lanes $lanes: "
  timeout 300 python3 bench.py --frame-batch 48 --frame-size 3840x2160 --frames-resident --lanes $lanes --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'MP/s', d['config']['frames_per_s'], 'frames/s', d['parity_gate'])"
done 2>&1 | tee gpurun_out/r06h/batch_lanes.txt
for sz in 16384 8192 4096 2048; do timeout 300 python tools/run_resident.py $sz 40 2>&1 | grep done | cut -c1-160; done | tee gpurun_out/r06h/resident.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "hot_path or golden or random_frames or values_outside or redo or redone or config3 or small_frames or one_column or corner or batch or error_behaviour" 2>&1 | tail -2
