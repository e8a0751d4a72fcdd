for x in 0 18432; do
echo "== extra LDS $x"
JXLT_TOKEN_EXTRA_LDS=$x JXLT_TRACE_EVENTS=1 timeout 300 python tools/run_resident.py 16384 6 2>&1 | grep -E "jxlt event|done" | tail -24
done
