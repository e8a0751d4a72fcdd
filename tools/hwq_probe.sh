# Does the number of hardware queues the runtime multiplexes the streams onto (GPU_MAX_HW_QUEUES, default 4) limit the
# frame batch (several device contexts of 6 streams each)?  And the single frame?
for q in ${QUEUES:-4 8 16}; do
for lanes in 3 6 8; do
  echo -n "GPU_MAX_HW_QUEUES=$q lanes $lanes: "
  GPU_MAX_HW_QUEUES=$q timeout 300 python3 bench.py --frame-batch 48 --frame-size 3840x2160 --frames-resident --lanes $lanes --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'MP/s', d['config']['frames_per_s'], 'frames/s')"
done
for sz in 2048 4096 8192 16384; do echo -n "GPU_MAX_HW_QUEUES=$q size $sz: "; GPU_MAX_HW_QUEUES=$q timeout 300 python tools/run_resident.py $sz 40 2>&1 | grep done | cut -c1-50; done
done
