#!/bin/bash
# Per-phase clocks of thread 0 in the production tile kernel (a -DJXLT_TIMING_MARKS build, prebuilt: gpurun_tmp/variants/0.so)
cp libjxl-tiny_amd/csrc/libjxltiny_hip.so gpurun_tmp/variants/production.so
cp gpurun_tmp/variants/0.so libjxl-tiny_amd/csrc/libjxltiny_hip.so
for sz in 8192 16384; do timeout 600 python tools/profile_phases.py $sz 2>&1 | head -16; done
cp gpurun_tmp/variants/production.so libjxl-tiny_amd/csrc/libjxltiny_hip.so
