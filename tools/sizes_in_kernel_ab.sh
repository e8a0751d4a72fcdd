timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rep in 1 2 3; do
for sz in 1024 2048 4096 8192; do
  for cfg in "JXLT_PACK_SIZES_IN_KERNEL=1" "JXLT_PACK_SIZES_IN_KERNEL=0"; do
    echo -n "== $sz [$cfg] "; env $cfg timeout 300 python tools/run_resident.py $sz 80 2>&1 | grep done | cut -c1-50
  done
done
done
for cfg in "JXLT_PACK_SIZES_IN_KERNEL=1" "JXLT_PACK_SIZES_IN_KERNEL=0"; do
echo -n "[$cfg] 4K batch 6 lanes: "
env $cfg timeout 300 python3 bench.py --frame-batch 48 --frame-size 3840x2160 --frames-resident --lanes 6 --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'MP/s', d['config']['frames_per_s'], 'frames/s')"
done
