JXLT_PACK_TWO_PASS=0 timeout 300 python tools/shard_overhead.py 16384 10 2>&1 | tail -4
timeout 300 python tools/shard_overhead.py 16384 10 2>&1 | tail -4
for sz in 16384 8192 4096 2048; do for cfg in "JXLT_PACK_TWO_PASS=0" "JXLT_PACK_TWO_PASS=1"; do echo -n "$sz [$cfg] "; env $cfg timeout 300 python tools/run_resident.py $sz 40 2>&1 | grep done | cut -c1-60; done; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
