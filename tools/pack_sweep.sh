#!/bin/bash
# Step time and the host's "head + place" time (from the AC sizes to the last byte in host memory, JXLT_TRACE) against
# the number of launches of the AC writing pass and the shape of their shares (experiment knobs of jxlt_capi.hip:
# JXLT_PACK_LAUNCHES, JXLT_PACK_GROWTH in percent: every share against the one before).  Usage: pack_sweep.sh "L:G ..."
cd "${GRAFT_REPO_ROOT:-.}"
run() { JXLT_TRACE=1 python bench.py --no-extras --steps 60 --warmup 10 > /tmp/ps.out 2> /tmp/ps.err; python3 - "$1" <<'PY'
import json, re, sys
d = json.loads(open('/tmp/ps.out').read().strip().splitlines()[-1])
v = [float(m.group(1)) for m in re.finditer(r"head \+ place ([0-9.]+)", open('/tmp/ps.err').read())][-60:]
m = [float(m.group(1)) for m in re.finditer(r"measure ([0-9.]+) \|", open('/tmp/ps.err').read())][-60:]
c = [float(m.group(1)) for m in re.finditer(r"codes ([0-9.]+) \|", open('/tmp/ps.err').read())][-60:]
mid = lambda a: sum(sorted(a)[len(a)//10:len(a)-len(a)//10]) / max(1, len(a) - 2 * (len(a)//10))
print(sys.argv[1], "step", d["ms_per_step"], "| AC code %.3f | measure %.3f | head+place %.3f (trimmed means)" % (mid(c), mid(m), mid(v)))
PY
}
for rep in 1 2; do
  for cfg in ${1:-3:200 5:100}; do
    IFS=: read L G <<< "$cfg"
    JXLT_PACK_LAUNCHES=$L JXLT_PACK_GROWTH=$G run "L:G=$cfg"
  done
done
