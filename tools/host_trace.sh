# Host-side stage times (JXLT_TRACE) and device events (JXLT_TRACE_EVENTS) of resident encodes of several sizes (GPU box).
for sz in 1024 2048 4096 8192; do
  echo "== $sz"
  JXLT_TRACE=1 timeout 300 python tools/run_resident.py $sz 8 2>&1 | grep "jxlt trace" | tail -4
  JXLT_TRACE_EVENTS=1 timeout 300 python tools/run_resident.py $sz 6 2>&1 | grep -E "jxlt event|done" | tail -18
done
