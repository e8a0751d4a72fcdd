#!/bin/bash
# the bench line of an unprofiled default run + the driver's command three times, with the round's committed profiles in place
mkdir -p gpurun_out/r06l
timeout 900 python -m pytest tests -m gpu -x -q -k "bench_self_launch or two_processes or over_two" 2>&1 | tail -3
timeout 300 python3 bench.py 2>gpurun_out/r06l/bench_default.err | tail -1 > gpurun_out/r06l/bench_line.json
for i in 1 2 3; do timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06l/driver_cmd_$i.json; done
python3 - <<'PY'
import json
for f in ["bench_line"] + ["driver_cmd_%d" % i for i in (1, 2, 3)]:
    d = json.loads(open("gpurun_out/r06l/%s.json" % f).read())
    print(f, d["value"], d["ms_per_step"], d.get("ms_per_step_median"), d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline"]["traffic_profile"]["stale"], d["roofline"].get("valu_issue", {}).get("profile", {}).get("stale"))
PY
cat gpurun_out/r06l/bench_default.err | tail -3
