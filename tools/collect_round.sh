#!/bin/bash
# The round's measurements in one GPU call: what tools/collect_profiles.sh writes (gpurun_out/prof) + tile_kernel's
# cycles / VALU per wave with the sources' fingerprint (gpurun_out/clk) + the driver's bench command three times.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
bash tools/collect_profiles.sh 16384 > gpurun_out/collect_profiles.log 2>&1
bash tools/tile_cycles.sh 16384 > gpurun_out/tile_cycles_16384.txt 2>&1
mkdir -p gpurun_out/prof
cp gpurun_out/clk/tile_valu_16384.json gpurun_out/prof/ 2>/dev/null
cp gpurun_out/tile_cycles_16384.txt gpurun_out/prof/
for i in 1 2 3; do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/prof/driver_cmd_$i.json
done
tail -5 gpurun_out/collect_profiles.log | cut -c1-300
cat gpurun_out/tile_cycles_16384.txt | tail -12
python3 - <<'PY'
import json
for i in (1, 2, 3):
    d = json.loads(open("gpurun_out/prof/driver_cmd_%d.json" % i).read())
    print(i, d["value"], d["ms_per_step"], d.get("ms_per_step_median"), d.get("ms_per_step_min"), d["roofline"]["frac"], d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
PY
