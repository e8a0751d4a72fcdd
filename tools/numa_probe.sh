#!/bin/bash
# Does the host's share of a step depend on which socket the encoding thread runs on?  bench.py (--no-extras) pinned
# to the CPUs next to GPU 0, to the other socket's, and unpinned.
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/numa; mkdir -p $OUT
: > $OUT/topology.txt
for d in /sys/bus/pci/devices/*; do
  if [ -f $d/local_cpulist ] && grep -qi "0x1002" $d/vendor 2>/dev/null && grep -q "^0x03\|^0x12" $d/class 2>/dev/null; then
    echo "$d class $(cat $d/class) numa $(cat $d/numa_node) cpus $(cat $d/local_cpulist)" >> $OUT/topology.txt
  fi
done
lscpu | grep -i "numa\|socket\|model name\|^cpu(s)" >> $OUT/topology.txt
echo "affinity of this shell: $(taskset -cp $$)" >> $OUT/topology.txt
NEAR=$(grep -m1 " cpus " $OUT/topology.txt | sed 's/.* cpus //')
echo "near: $NEAR" >> $OUT/topology.txt
ALL=$(cat /sys/devices/system/cpu/online)
FAR=$(python3 - "$NEAR" "$ALL" <<'PY'
import sys
def parse(s):
    out=set()
    for part in s.strip().split(","):
        if not part: continue
        a,_,b=part.partition("-")
        out.update(range(int(a), int(b or a)+1))
    return out
far=sorted(parse(sys.argv[2])-parse(sys.argv[1]))
print(",".join(map(str,far)))
PY
)
echo "far: $(echo $FAR | cut -c1-80)..." >> $OUT/topology.txt
for rep in 1 2; do
  python3 bench.py --steps 20 --warmup 5 --no-extras > $OUT/unpinned_$rep.json 2>/dev/null
  taskset -c "$NEAR" python3 bench.py --steps 20 --warmup 5 --no-extras > $OUT/near_$rep.json 2>/dev/null
  if [ -n "$FAR" ]; then taskset -c "$FAR" python3 bench.py --steps 20 --warmup 5 --no-extras > $OUT/far_$rep.json 2>/dev/null; fi
done
cat $OUT/topology.txt
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/numa/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d["kernel_ms"]
        print("%-28s step %.3f median %.3f min %.3f  tile %.3f tok %.3f  rest %.3f" % (f.split("/")[-1], d["ms_per_step"], d["ms_per_step_median"], d["ms_per_step_min"], k["tile_kernel"], k["tokenisation_after_tile_kernel"], d["ms_per_step"] - k["tile_kernel"] - k["tokenisation_after_tile_kernel"]))
    except Exception as e:
        print(f, "ERR", e)
PY
