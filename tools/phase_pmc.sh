#!/bin/bash
# Per-phase counters of tile_kernel: a -DJXLT_PHASE_STOPS build truncated after each phase, one rocprofv3 pass
# per counter set.  Usage: phase_pmc.sh <size> '<counter set 1>' '<counter set 2>' ...
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
SIZE=$1; shift
touch libjxl-tiny_amd/csrc/jxlt_device_common.h
make -C libjxl-tiny_amd -s -j3 csrc/libjxltiny_hip.so HIPFLAGS_EXTRA="-DJXLT_PHASE_STOPS" 2>&1 | grep -i error
rm -rf gpurun_out/phase_pmc
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/phase_pmc/$i -- python3 tools/phase_pmc.py run $SIZE > gpurun_out/phase_pmc_$i.log 2>&1
  python3 tools/phase_pmc.py report gpurun_out/phase_pmc/$i
done
