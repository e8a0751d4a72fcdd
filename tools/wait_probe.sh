#!/bin/bash
# Does the host's wake-up latency behind hipEventSynchronize / hipStreamSynchronize cost the step anything?
# Alternates the default (interrupt) wait with the runtime's active wait (ROC_ACTIVE_WAIT_TIMEOUT, microseconds).
cd "${GRAFT_REPO_ROOT:-.}"
run() { python bench.py --no-extras --steps 40 --warmup 10 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['kernel_ms'])"; }
for rep in 1 2 3; do
  run default
  ROC_ACTIVE_WAIT_TIMEOUT=5000 run active
done
