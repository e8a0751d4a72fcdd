JXLT_TRACE=1 JXLT_TRACE_EVENTS=1 timeout 300 python tools/run_resident.py 16384 10 2>&1 | tail -45
