export JXLT_BENCH_ONE_DEVICE=1
for cfg in "JXLT_WAIT_SLEEP=0" "JXLT_WAIT_SLEEP=1"; do
  echo "== 8 ranks [$cfg]"
  cat /sys/fs/cgroup/cpu.stat | grep -E "nr_throttled|throttled_usec"
  env $cfg timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 8 --steps 5 --warmup 2 --no-extras 2>/dev/null | grep "^{" | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['n_gpus'], d['value'], d['ms_per_step'], d.get('step_ms'))"
  cat /sys/fs/cgroup/cpu.stat | grep -E "nr_throttled|throttled_usec"
done
