#!/bin/bash
# A/B of kernel build variants on the GPU box: rebuild the HIP library with extra flags (argument 1: a list of
# flag sets separated by ';'), report kernel times of a device-only bench run for each.
cd "${GRAFT_REPO_ROOT:-.}"
IFS=';' read -ra VARIANTS <<< "$1"
for v in "${VARIANTS[@]}"; do
  touch libjxl-tiny_amd/csrc/jxlt_capi.hip
  make -C libjxl-tiny_amd -s csrc/libjxltiny_hip.so HIPFLAGS_EXTRA="$v" 2>&1 | grep -i error
  timeout 200 python bench.py --no-extras --steps 8 2>&1 | tail -1 | V="$v" python3 -c "
import json,sys,os
d=json.loads(sys.stdin.readline()); print('[%s]' % os.environ['V'], d['ms_per_step'], d['kernel_ms'])"
done
