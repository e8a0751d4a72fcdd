#!/bin/bash
# Same-box A/B of two source trees (see ab_trees.sh) by the time per complete resident encode at several frame sizes.
ROOT="${GRAFT_REPO_ROOT:-$PWD}"
OTHER="$ROOT/${1:-ab_prev}"
for rep in $(seq 1 ${REPS:-2}); do
  for sz in ${SIZES:-2048 4096 8192}; do
    echo -n "this  $sz: "; (cd "$ROOT" && timeout 300 python3 tools/run_resident.py $sz ${PASSES:-120} 2>&1 | grep done | sed -e "s#{.*}##")
    echo -n "other $sz: "; (cd "$OTHER" && timeout 300 python3 tools/run_resident.py $sz ${PASSES:-120} 2>&1 | grep done | sed -e "s#{.*}##")
  done
done
