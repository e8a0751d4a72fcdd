#!/bin/bash
# Host time of the code constructions inside complete encodes (JXLT_TRACE): the clustering's own choice between
# sharing its Huffman costs with the helper threads and working alone ("auto") against both forced ways
# (JXLT_POOL_MODE=1 / 0: experiment knob of host/entropy_coder.cc).  Usage: code_probe.sh "<modes>" [distance]
cd "${GRAFT_REPO_ROOT:-.}"
run() { JXLT_TRACE=1 python bench.py --no-extras --steps 40 --warmup 10 --distance ${DIST:-1} > /tmp/ps.out 2> /tmp/ps.err; python3 - "$1" <<'PY'
import json, re, sys
d = json.loads(open('/tmp/ps.out').read().strip().splitlines()[-1])
err = open('/tmp/ps.err').read()
mid = lambda a: sum(sorted(a)[len(a)//10:len(a)-len(a)//10]) / max(1, len(a) - 2 * (len(a)//10))
c = [float(m.group(1)) for m in re.finditer(r"\| ac code ([0-9.]+)", err)][-40:]
dc = [float(m.group(1)) for m in re.finditer(r"dc code ([0-9.]+)", err)][-40:]
print(sys.argv[1], "step", d["ms_per_step"], "| AC code %.3f ms | DC code %.3f ms (trimmed means)" % (mid(c), mid(dc)))
PY
}
DIST=${2:-1}
for rep in ${REPS:-1 2 3}; do
  for t in ${1:-auto 0 1}; do
    if [ "$t" = auto ]; then unset JXLT_POOL_MODE; else export JXLT_POOL_MODE=$t; fi
    run "mode=$t"
  done
done
grep "clustering" /tmp/ps.err | sort | uniq -c | sort -rn | head -4
