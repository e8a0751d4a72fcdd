#!/bin/bash
# Round 6: same-box A/B against ab_prev (the round's start), then probe builds of P9 (parts left out: timing only).
mkdir -p gpurun_out/r06d
REPS=2 bash tools/ab_trees.sh 2>&1 | tee gpurun_out/r06d/ab_trees.txt
bash tools/ab_cycles.sh "base;-DJXLT_ABL_P9=1;-DJXLT_ABL_P9=2;-DJXLT_ABL_P9=4;-DJXLT_ABL_P9=7;-DJXLT_ABL_P9=8;base" 2>&1 | tee gpurun_out/r06d/abl_p9.txt
