#!/bin/bash
mkdir -p gpurun_out/r06e
REPS=2 bash tools/run_variants.sh 2>&1 | tee gpurun_out/r06e/probes_p9.txt
timeout 600 python tools/slab_of_8.py 16384 30 gpurun_out/r06e/slab_of_8.json 2>&1 | tail -3 | cut -c1-1500
