#!/bin/bash
# closing measurements with the round's final build: bench lines (profiles in place), batch lanes, small frames, one rank of eight
mkdir -p gpurun_out/r06z
timeout 300 python3 bench.py 2>gpurun_out/r06z/bench_default.err | tail -1 > gpurun_out/r06z/bench_line.json
for i in 1 2 3; do timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06z/driver_cmd_$i.json; done
python3 - <<'PY'
import json
for f in ["bench_line"] + ["driver_cmd_%d" % i for i in (1, 2, 3)]:
    d = json.loads(open("gpurun_out/r06z/%s.json" % f).read())
    print(f, d["value"], d["ms_per_step"], d.get("ms_per_step_median"), d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline"]["traffic_profile"]["stale"])
PY
for rep in 1 2; do
for lanes in 1 2 4 6 8; do
  echo -n "lanes $lanes: "
  timeout 300 python3 bench.py --frame-batch 48 --frame-size 3840x2160 --frames-resident --lanes $lanes --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'MP/s', d['config']['frames_per_s'], 'frames/s', d['parity_gate'])"
done
done 2>&1 | tee gpurun_out/r06z/batch_lanes.txt
for sz in 8192 4096 2048 1024; do timeout 300 python tools/run_resident.py $sz 300 2>&1 | grep done | sed -e 's/{.*}//'; done | tee gpurun_out/r06z/resident_small.txt
timeout 600 python3 tools/slab_of_8.py > gpurun_out/r06z/slab_of_8.log 2>&1; tail -3 gpurun_out/r06z/slab_of_8.log | cut -c1-300
