#!/bin/bash
mkdir -p gpurun_out/r06g
REPS=2 bash tools/run_variants.sh 2>&1 | tee gpurun_out/r06g/probes_dephase.txt
