#!/bin/bash
# Host time of the code constructions inside complete encodes (JXLT_TRACE) against the number of non-zero counts
# from which the clustering shares its Huffman costs with the helper threads (JXLT_POOL_MIN_SYMBOLS: experiment
# knob of host/entropy_coder.cc; 0 = always shared, a huge number = never).
cd "${GRAFT_REPO_ROOT:-.}"
run() { JXLT_TRACE=1 python bench.py --no-extras --steps 60 --warmup 10 > /tmp/ps.out 2> /tmp/ps.err; python3 - "$1" <<'PY'
import json, re, sys
d = json.loads(open('/tmp/ps.out').read().strip().splitlines()[-1])
err = open('/tmp/ps.err').read()
mid = lambda a: sum(sorted(a)[len(a)//10:len(a)-len(a)//10]) / max(1, len(a) - 2 * (len(a)//10))
c = [float(m.group(1)) for m in re.finditer(r"\| ac code ([0-9.]+)", err)][-60:]
dc = [float(m.group(1)) for m in re.finditer(r"dc code ([0-9.]+)", err)][-60:]
print(sys.argv[1], "step", d["ms_per_step"], "| AC code %.3f ms | DC code %.3f ms (trimmed means)" % (mid(c), mid(dc)))
PY
}
grep -m2 "clustering" /tmp/ps.err 2>/dev/null
for rep in 1 2 3; do
  for t in ${1:-450 0 100000}; do JXLT_POOL_MIN_SYMBOLS=$t run "min_symbols=$t"; done
done
grep "clustering" /tmp/ps.err | sort | uniq -c | sort -rn | head -4
