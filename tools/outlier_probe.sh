# Where do the occasional 12 ms steps of the driver's bench command come from?  Alternating runs with and without the
# CPU's first touch of large page-locked buffers: the slow steps (> 5.6 ms) and the warm-up steps of each.
for i in 1 2 3 4 5 6 7 8; do
for cfg in ${CFGS:-"JXLT_PINNED_TOUCH=1" "JXLT_PINNED_TOUCH=0"}; do
  echo -n "[$cfg] "
  env $cfg timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'], [s for s in d['step_ms'] if s > 5.6], d['warmup_step_ms'])"
done
done
grep -E "nr_throttled" /sys/fs/cgroup/cpu.stat
