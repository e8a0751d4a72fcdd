#!/bin/bash
# A/B of the 8-wave and the 12-wave variant of tile_kernel on the GPU box (JXLT_TILE_WAVES): quick parity tests
# (PARITY=1), shader cycles / VALU per wave (tools/tile_cycles.sh), kernel times of a bench run without extras.
cd "${GRAFT_REPO_ROOT:-.}"
for w in ${WAVES:-8 12}; do
  export JXLT_TILE_WAVES=$w
  echo "== JXLT_TILE_WAVES=$w"
  if [ -n "$PARITY" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "hot_path or golden or random or values_outside" 2>&1 | tail -2; fi
  ./tools/tile_cycles.sh ${SIZE:-16384} | grep -E "tile|token_kernel"
  timeout 300 python bench.py --no-extras --steps 8 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms'])"
done
