JXLT_TRACE=1 JXLT_TRACE_EVENTS=1 timeout 300 python tools/run_resident.py 8192 12 2>&1 | tail -26
bash tools/ranks_on_one_gpu.sh
