# Sweep of the single pass's launch shares (GPU box): JXLT_PACK_LAUNCHES x JXLT_PACK_GROWTH, time per resident encode.
mkdir -p gpurun_out/r04s
: > gpurun_out/r04s/sweep.log
for sz in 16384 8192; do
for l in 3 4 5 6 8; do
for g in 100 140 200; do
  echo -n "size $sz launches $l growth $g: " >> gpurun_out/r04s/sweep.log
  JXLT_PACK_LAUNCHES=$l JXLT_PACK_GROWTH=$g timeout 300 python tools/run_resident.py $sz 40 2>&1 | grep done >> gpurun_out/r04s/sweep.log
done
done
echo -n "size $sz two-pass: " >> gpurun_out/r04s/sweep.log
JXLT_PACK_TWO_PASS=1 timeout 300 python tools/run_resident.py $sz 40 2>&1 | grep done >> gpurun_out/r04s/sweep.log
done
cat gpurun_out/r04s/sweep.log
