#!/bin/bash
# The round's closing checks on the GPU box: random C-ABI call sequences and a random parity sweep against the oracle
# (default packing form, and two passes forced), then the GPU suite.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/final
timeout 1500 python3 tools/api_fuzz.py 100 300 > gpurun_out/final/api_fuzz_100x300.log 2>&1; echo "api_fuzz rc=$?"
JXLT_PACK_TWO_PASS=1 timeout 900 python3 tools/api_fuzz.py 30 300 > gpurun_out/final/api_fuzz_two_pass_30x300.log 2>&1; echo "api_fuzz two-pass rc=$?"
timeout 2400 python3 tools/gpu_sweep.py ${SWEEP:-400} > gpurun_out/final/gpu_sweep${SWEEP:-400}.log 2>&1; echo "gpu_sweep rc=$?"
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_gpu.log 2>&1; echo "pytest rc=$?"
for f in gpurun_out/final/*.log; do tail -n 2 $f; done #/final/api_fuzz_100x300.log gpurun_out/final/api_fuzz_two_pass_30x300.log gpurun_out/final/gpu_sweep400.log gpurun_out/final/pytest_gpu.log
