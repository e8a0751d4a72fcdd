#!/bin/bash
# A/B of kernel build variants on the GPU box: rebuild the HIP library with extra flags (argument 1: a list of
# flag sets separated by ';'; an empty set = the production build), then for each: tile_cycles (shader cycles,
# VALU per wave) and the kernel times of a bench run without extras.  PARITY=1 also runs the quick GPU parity tests,
# PACK=1 the counters of the section-packing kernels (tools/pack_cycles.sh).
cd "${GRAFT_REPO_ROOT:-.}"
IFS=';' read -ra VARIANTS <<< "$1"
for v in "${VARIANTS[@]}"; do
  [ "$v" = "base" ] && v=""
  touch libjxl-tiny_amd/csrc/jxlt_device_common.h
  make -C libjxl-tiny_amd -s -j3 csrc/libjxltiny_hip.so HIPFLAGS_EXTRA="$v" 2>&1 | grep -i error
  echo "== [$v]"
  if [ -n "$PARITY" ]; then timeout 600 python -m pytest tests -m gpu -x -q -k "hot_path or golden or random or values_outside" 2>&1 | tail -1; fi
  ./tools/tile_cycles.sh 16384 | grep -E "tile(12)?_kernel|token_kernel"
  if [ -n "$PACK" ]; then ./tools/pack_cycles.sh 16384 pack_tile_; fi
  timeout 200 python bench.py --no-extras --steps 8 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms'])"
done
