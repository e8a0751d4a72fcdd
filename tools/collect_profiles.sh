#!/bin/bash
# Regenerates what profiles/ holds, on the GPU box:
#   gpurun_out/prof/bench_stats      rocprofv3 --kernel-trace --stats of the default bench.py run
#   gpurun_out/prof/traffic_<size>.json  HBM bytes per launch (separate FETCH_SIZE / WRITE_SIZE passes + calibration)
#   gpurun_out/prof/bench_line.json  the bench line of an unprofiled run
# Usage: tools/collect_profiles.sh [size]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
SIZE=${1:-16384}
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
# (--no-extras: only the timed workload's launches, so that the per-kernel averages are those of the 16384^2 frame;
#  the default run's extra legs launch the same kernels on crops and row slabs)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- python3 bench.py --no-extras --steps 10 > $OUT/bench_profiled.log 2>&1
grep "^{\"metric\"" $OUT/bench_profiled.log > $OUT/bench_line_profiled.json
timeout 300 python3 bench.py 2>/dev/null | tail -1 > $OUT/bench_line.json
if [ ! -x tools/fetch_calib ]; then hipcc --offload-arch=gfx950 -O2 -o tools/fetch_calib tools/fetch_calib.hip; fi
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/calib_$c -- ./tools/fetch_calib > $OUT/calib_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/enc_$c -- python3 tools/run_resident.py $SIZE 3 > $OUT/enc_$c.log 2>&1
done
f() { find $OUT/$1 -name '*counter_collection.csv' | head -1; }
python3 tools/collect_traffic.py "$(f calib_FETCH_SIZE)" "$(f calib_WRITE_SIZE)" "$(f enc_FETCH_SIZE)" "$(f enc_WRITE_SIZE)" $SIZE $OUT/traffic_$SIZE.json 3
./tools/pack_cycles.sh $SIZE jxlt_dev > $OUT/kernel_cycles_$SIZE.txt 2>&1
./tools/pmc_util.sh 8192 > $OUT/tile8192_valu_util.txt 2>&1
python3 tools/config_table.py 2>/dev/null > $OUT/config_table.jsonl
find $OUT/bench_stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -8 $OUT/kernel_stats.csv | cut -c1-160
cat $OUT/bench_line.json | cut -c1-600
