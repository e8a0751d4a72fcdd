#!/bin/bash
# Round 6: the GPU suite with the current build, then the wave-priority A/B of tile12_kernel (cycles).
mkdir -p gpurun_out/r06b
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06b/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r06b/pytest.log
REPS=1 bash tools/ab_cycles.sh "base;-DJXLT_PRIO_P9=1;-DJXLT_PRIO_P9=2;-DJXLT_PRIO_SERIAL=2;-DJXLT_PRIO_P9=2 -DJXLT_PRIO_SERIAL=3;base" 2>&1 | tee gpurun_out/r06b/ab_prio.txt
