# Event trace (and, second run, look-back statistics) of the single-pass packing (GPU box).
mkdir -p gpurun_out/r04s
for sz in 16384 4096; do
JXLT_TRACE_EVENTS=1 timeout 300 python tools/run_resident.py $sz 6 2>&1 | grep -E "look-back|stream launch|token_kernel done|copy|done" | tail -17
JXLT_TRACE_EVENTS=1 JXLT_LOOKBACK_STATS=1 timeout 300 python tools/run_resident.py $sz 6 2>&1 | grep -E "look-back" | tail -2
done
