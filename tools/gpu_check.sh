#!/bin/bash
# One GPU-box round trip: parity tests, per-phase instruction counts, bench line.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
./tools/tile_cycles.sh 16384 | grep jxlt_dev
if [ -n "$PHASES" ]; then
# profiling build with the phase stops (the box copy is scratch; the production .so is rebuilt below)
cp libjxl-tiny_amd/csrc/libjxltiny_hip.so /tmp/prod_hip.so
touch libjxl-tiny_amd/csrc/jxlt_device_common.h
make -C libjxl-tiny_amd -s -j3 csrc/libjxltiny_hip.so HIPFLAGS_EXTRA=-DJXLT_PHASE_STOPS 2>&1 | grep -i error
rm -rf gpurun_out/phase_pmc
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv \
  -d gpurun_out/phase_pmc -- python3 tools/phase_pmc.py run 4096 > gpurun_out/phase_pmc.log 2>&1
python3 tools/phase_pmc.py report gpurun_out/phase_pmc
timeout 200 python3 tools/profile_phases.py 8192 2>&1 | grep -E "cycles/tile|stop after"
cp /tmp/prod_hip.so libjxl-tiny_amd/csrc/libjxltiny_hip.so
fi
timeout 300 python bench.py "$@" 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('value %.1f %s  ms/step %.3f  kernels %s  parity %s' % (d['value'], d['unit'], d['ms_per_step'], d['kernel_ms'], d['parity_gate']))"
