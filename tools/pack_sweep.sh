#!/bin/bash
# Step time against the number of launches of the AC writing pass and the shape of their shares
# (JXLT_PACK_LAUNCHES, JXLT_PACK_SHRINK: experiment knobs of jxlt_capi.hip).  Usage: pack_sweep.sh "L:S L:S ..."
cd "${GRAFT_REPO_ROOT:-.}"
run() { python bench.py --no-extras --steps 60 --warmup 10 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'])"; }
for rep in 1 2 3; do
  for cfg in ${1:-5:40 8:40 5:80}; do JXLT_PACK_LAUNCHES=${cfg%%:*} JXLT_PACK_SHRINK=${cfg##*:} run "L:S=$cfg"; done
done
