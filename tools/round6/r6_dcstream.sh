#!/bin/bash
# DC-group sections' packing on its own stream for small frames? (JXLT_DC_PACK_STREAM knob)
for rep in 1 2 3; do
for k in 0 1; do
for sz in 4096 2048 1024; do echo -n "JXLT_DC_PACK_STREAM=$k "; JXLT_DC_PACK_STREAM=$k timeout 300 python tools/run_resident.py $sz 600 2>&1 | grep done | sed -e 's/{.*}//' ; done
done
done
