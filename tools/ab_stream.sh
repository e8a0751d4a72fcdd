# A/B of the packing stage (GPU box): single pass (default) against measure + write (JXLT_PACK_TWO_PASS=1), and the
# DC-group sections packed on their own stream (default) against the main stream (JXLT_DC_PACK_STREAM=0).
mkdir -p gpurun_out/r04s
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04s/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r04s/rc.txt
: > gpurun_out/r04s/ab.log
for rep in 1 2; do
for sz in 16384 8192 4096 2048; do
  for cfg in "" "JXLT_PACK_TWO_PASS=1" "JXLT_DC_PACK_STREAM=0" "JXLT_PACK_TWO_PASS=1 JXLT_DC_PACK_STREAM=0"; do
    echo -n "== $sz [$cfg] " >> gpurun_out/r04s/ab.log
    env $cfg timeout 300 python tools/run_resident.py $sz 40 2>&1 | grep done >> gpurun_out/r04s/ab.log
  done
done
done
JXLT_TRACE_EVENTS=1 timeout 300 python tools/run_resident.py 16384 6 > gpurun_out/r04s/trace16384.log 2>&1
tail -3 gpurun_out/r04s/pytest.log; cut -c1-100 gpurun_out/r04s/ab.log; grep "jxlt event" gpurun_out/r04s/trace16384.log | tail -17
