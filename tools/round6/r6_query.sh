#!/bin/bash
for rep in 1 2 3; do
for kb in 0 2048 16384; do
  echo -n "limit $kb KB 8192: "; JXLT_DELIVER_QUERY_KB=$kb timeout 300 python tools/run_resident.py 8192 300 2>&1 | grep done | sed -e 's/{.*}//'
done
done
timeout 900 python -m pytest tests -m gpu -x -q -k "hot_path or config or random_frames or sequences or error_behaviour or every_form or pfm" 2>&1 | tail -2
