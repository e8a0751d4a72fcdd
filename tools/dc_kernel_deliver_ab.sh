for rep in 1 2 3; do
for cfg in "JXLT_DC_DELIVER_KERNEL=0" "JXLT_DC_DELIVER_KERNEL=1"; do echo -n "16384 [$cfg] "; env $cfg timeout 300 python tools/run_resident.py 16384 40 2>&1 | grep done | cut -c1-60; done
done
JXLT_DC_DELIVER_KERNEL=1 JXLT_TRACE_EVENTS=1 timeout 300 python tools/run_resident.py 16384 6 2>&1 | grep -E "jxlt event" | tail -18
