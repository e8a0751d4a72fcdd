#!/bin/bash
# same-box A/B of the resident 3840x2160 batch: this tree against ab_prev/
ROOT="${GRAFT_REPO_ROOT:-$PWD}"
run() { (cd "$1" && timeout 300 python3 bench.py --frame-batch 48 --frame-size 3840x2160 --frames-resident --lanes $2 --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'MP/s')"); }
for rep in 1 2; do
for lanes in 2 4 6; do
  echo -n "this  lanes $lanes: "; run "$ROOT" $lanes
  echo -n "other lanes $lanes: "; run "$ROOT/ab_prev" $lanes
done
done
