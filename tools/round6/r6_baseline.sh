#!/bin/bash
# Round 6, first GPU call: the whole GPU suite, the clock-independent cycle counts, the driver's bench command.
mkdir -p gpurun_out/r06a
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06a/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r06a/pytest.log
bash tools/tile_cycles.sh 16384 | tee gpurun_out/r06a/tile_cycles.txt
cp gpurun_out/clk/tile_valu_16384.json gpurun_out/r06a/ 2>/dev/null
for sz in 16384 8192 4096 2048; do timeout 300 python tools/run_resident.py $sz 40 2>&1 | grep done | cut -c1-160; done | tee gpurun_out/r06a/resident.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06a/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_median','ms_per_step_min')}, d['roofline'], d.get('harness'))
PY
