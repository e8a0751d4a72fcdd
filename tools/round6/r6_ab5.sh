#!/bin/bash
mkdir -p gpurun_out/r06f
REPS=2 bash tools/run_variants.sh 2>&1 | tee gpurun_out/r06f/probes_phases.txt
