#!/bin/bash
# Host + device stage trace of small resident frames (where does a 4096^2 encode spend its 0.53 ms?)
mkdir -p gpurun_out/r06t
for sz in 4096 2048; do
  JXLT_TRACE=1 JXLT_TRACE_EVENTS=1 timeout 300 python tools/run_resident.py $sz 12 > gpurun_out/r06t/trace_$sz.txt 2>&1
  tail -60 gpurun_out/r06t/trace_$sz.txt
done
