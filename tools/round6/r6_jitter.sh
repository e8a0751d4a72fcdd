#!/bin/bash
for rep in 1 2; do
for m in default 0 1; do
  echo "== JXLT_POOL_MODE=$m"
  if [ $m = default ]; then timeout 300 python3 tools/stage_jitter.py 16384 200; else JXLT_POOL_MODE=$m timeout 300 python3 tools/stage_jitter.py 16384 200; fi
done
done 2>&1 | grep -v "amdgpu.ids"
