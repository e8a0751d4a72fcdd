# A/B for small and medium resident frames (GPU box): DC-group sections on their own stream or not.
for rep in 1 2 3; do
for sz in 1024 2048 4096 8192; do
  for cfg in "JXLT_DC_PACK_STREAM=0" "JXLT_DC_PACK_STREAM=1"; do
    echo -n "== $sz [$cfg] "
    env $cfg timeout 300 python tools/run_resident.py $sz 120 2>&1 | grep done | cut -c1-60
  done
done
done
