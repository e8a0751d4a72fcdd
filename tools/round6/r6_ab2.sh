#!/bin/bash
# Round 6: quick GPU parity tests with the current build, then the same-box cycle A/B against ab_prev/ (the round's baseline).
mkdir -p gpurun_out/r06c
timeout 1200 python -m pytest tests -m gpu -x -q -k "hot_path or golden or random_frames or values_outside or redo or redone or config3 or small_frames or one_column or corner" > gpurun_out/r06c/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r06c/pytest.log
REPS=${REPS:-3} bash tools/ab_trees.sh 2>&1 | tee gpurun_out/r06c/ab_trees.txt
