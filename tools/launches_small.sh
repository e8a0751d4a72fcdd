# Number of launches of the AC sections' single pass for small frames (GPU box).
for l in 1 2 3; do
for sz in 1024 2048 4096 8192; do
  echo -n "launches $l size $sz: "; JXLT_PACK_LAUNCHES=$l timeout 300 python tools/run_resident.py $sz 60 2>&1 | grep done | cut -c1-50
done
echo -n "launches $l 4K batch, 6 lanes: "
JXLT_PACK_LAUNCHES=$l timeout 300 python3 bench.py --frame-batch 48 --frame-size 3840x2160 --frames-resident --lanes 6 --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'MP/s', d['config']['frames_per_s'], 'frames/s')"
done
