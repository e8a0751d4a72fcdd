CFGS="JXLT_COPY_WARMUP=3" bash tools/outlier_probe.sh 2>&1 | head -8
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 300 python3 tools/soak.py 4096 500 2>&1 | grep "contexts x" | head -3
for i in 1 2 3; do timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_median'], max(d['step_ms']), d['warmup_step_ms'])"; done
