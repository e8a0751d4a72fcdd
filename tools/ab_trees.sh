#!/bin/bash
# Same-box A/B of two source trees: the repository itself against a second checkout inside it (default ab_prev/, made
# with `git archive <rev> | tar -x -C ab_prev` and built before the GPU call).  tile12_kernel's shader cycles per
# 16384^2 launch, alternating, REPS times each (the pool's boxes differ by ~1 % in cycles: only runs on one box compare).
ROOT="${GRAFT_REPO_ROOT:-$PWD}"
OTHER="$ROOT/${1:-ab_prev}"
mkdir -p "$OTHER/gpurun_out"
SIZE=${SIZE:-16384}
for rep in $(seq 1 ${REPS:-3}); do
  echo -n "this  "; GRAFT_REPO_ROOT="$ROOT" "$ROOT/tools/tile_cycles.sh" $SIZE | grep -E "tile12_kernel  "
  echo -n "other "; GRAFT_REPO_ROOT="$OTHER" "$OTHER/tools/tile_cycles.sh" $SIZE | grep -E "tile12_kernel  "
done
