#!/bin/bash
# Round 6, closing call 1: the whole GPU suite, then the round's profiles (tools/collect_round.sh).
mkdir -p gpurun_out/r06final
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06final/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r06final/pytest_gpu.log
bash tools/collect_round.sh 2>&1 | tail -30
