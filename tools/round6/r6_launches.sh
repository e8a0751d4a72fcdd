#!/bin/bash
for rep in 1 2; do
for n in 1 2 3; do
for sz in 8192 4096; do
  echo -n "JXLT_PACK_LAUNCHES=$n $sz: "; JXLT_PACK_LAUNCHES=$n timeout 300 python tools/run_resident.py $sz 300 2>&1 | grep done | sed -e 's/{.*}//'
done
done
done
