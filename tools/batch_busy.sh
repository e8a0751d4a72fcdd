#!/bin/bash
# How busy is the GPU during a resident 3840x2160 batch?  Kernel trace of `bench.py --frame-batch ... --frames-resident`:
# the union of all kernel intervals against the span they cover, and the busy time per kernel.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
LANES=${1:-6}
rm -rf gpurun_out/bb
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bb -- python3 bench.py --frame-batch 48 --frame-size 3840x2160 --frames-resident --lanes $LANES --steps 6 --warmup 2 > gpurun_out/bb.log 2>&1
tail -1 gpurun_out/bb.log | cut -c1-200
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/bb/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-34:]) for r in rows)
# the last 60 % of the trace = steady state
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * 4 // 10
ev = [e for e in ev if e[0] >= t_lo]
span = ev[-1][1] - ev[0][0]
busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
for s, e, _ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
per = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev:
    per[n][0] += e - s
    per[n][1] += 1
frames = per["jxlt_dev::tile12_kernel"][1]
print("steady state: %.1f ms, %d frames, %.3f ms per frame; some kernel running %.1f %% of the time" % (span / 1e6, frames, span / 1e6 / max(1, frames), 100.0 * busy / span))
tot = sum(v[0] for v in per.values())
print("sum of kernel durations per frame %.3f ms (overlapping kernels counted each)" % (tot / 1e6 / max(1, frames)))
for n, (d, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %-36s %6.1f us per frame  (%d launches, %.1f us each)" % (n, d / 1e3 / max(1, frames), c, d / 1e3 / c))
PY
