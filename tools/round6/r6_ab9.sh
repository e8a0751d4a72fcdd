#!/bin/bash
mkdir -p gpurun_out/r06j
timeout 1200 python -m pytest tests -m gpu -x -q -k "hot_path or golden or random_frames or values_outside or redo or redone or config3 or small_frames or one_column or corner" 2>&1 | tail -2
REPS=3 bash tools/ab_trees.sh 2>&1 | tee gpurun_out/r06j/ab_trees.txt
