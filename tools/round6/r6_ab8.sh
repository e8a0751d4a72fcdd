#!/bin/bash
mkdir -p gpurun_out/r06i
REPS=2 bash tools/run_variants.sh 2>&1 | tee gpurun_out/r06i/probes_tiles_per_wg.txt
