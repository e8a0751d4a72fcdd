JXLT_PACK_TWO_PASS=0 JXLT_PACK_LAUNCHES=4 JXLT_PACK_GROWTH=100 JXLT_TRACE_EVENTS=1 timeout 300 python tools/run_resident.py 16384 6 2>&1 | grep -E "jxlt event" | tail -22
