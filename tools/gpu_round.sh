mkdir -p gpurun_out/r04s
timeout 2000 python -m pytest tests -m gpu -x -q > gpurun_out/r04s/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r04s/pytest.log
for sz in 16384 8192 4096 2048 1024; do timeout 300 python tools/run_resident.py $sz 40 2>&1 | grep done | cut -c1-120; done
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04s/bench.json 2> gpurun_out/r04s/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04s/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_median','ms_per_step_min')}, d['roofline'])
PY
