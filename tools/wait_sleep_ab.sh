for rep in 1 2 3; do
for sz in 16384 8192; do
for cfg in "JXLT_WAIT_SLEEP=1" "JXLT_WAIT_SLEEP=0"; do
echo -n "$sz [$cfg] "; env $cfg timeout 300 python tools/run_resident.py $sz 60 2>&1 | grep done | cut -c1-55
done; done; done
for cfg in "JXLT_WAIT_SLEEP=1" "JXLT_WAIT_SLEEP=0"; do
echo -n "[$cfg] 4K batch 8 lanes: "
env $cfg timeout 300 python3 bench.py --frame-batch 48 --frame-size 3840x2160 --frames-resident --lanes 8 --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'MP/s', d['config']['frames_per_s'], 'frames/s')"
echo -n "[$cfg] driver cmd: "
( time env $cfg timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min'])" ) 2>&1 | grep -E "^[0-9]|user|sys" | tr '\n' ' '; echo
done
