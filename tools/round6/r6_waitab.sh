#!/bin/bash
for rep in 1 2 3; do
for cfg in "JXLT_X=1" "JXLT_WAIT_NO_SLEEP=1"; do
  echo -n "[$cfg] "; env $cfg timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min'], d['step_diagnostics']['host_stage_ms_median'])"
done
done
for cfg in "JXLT_X=1" "JXLT_WAIT_NO_SLEEP=1"; do for sz in 4096 2048; do echo -n "[$cfg] "; env $cfg timeout 300 python tools/run_resident.py $sz 60 2>&1 | grep done | cut -c1-60; done; done
