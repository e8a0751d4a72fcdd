cd "${GRAFT_REPO_ROOT:-.}"
# (round 3 compared the 8-wave and the 12-wave kernel with this; the 8-wave kernel went in round 4)
for rep in 1 2; do for w in 12; do
  ( for i in 1 2 3 4 5 6; do sleep 1.5; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo; done ) > gpurun_out/smi_$w_$rep.log 2>&1 &
  SMI=$!
  timeout 300 python bench.py --no-extras --steps 1500 --warmup 20 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('waves $w rep $rep', d['ms_per_step'], d['kernel_ms'])"
  wait $SMI
  cat gpurun_out/smi_$w_$rep.log | tail -4
done; done
