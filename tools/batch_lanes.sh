# Resident 3840x2160 frame batches over 1..6 lanes (device contexts) on one GPU: frames per second and Mpixels/s
# (">= 30 GP/s and monotone in lanes up to 6").  JXLT_PACK_TWO_PASS=1 beside it for comparison.
for cfg in "" "JXLT_PACK_TWO_PASS=1"; do
for lanes in 1 2 3 4 6; do
  echo -n "[$cfg] lanes $lanes: "
  env $cfg timeout 300 python3 bench.py --frame-batch 48 --frame-size 3840x2160 --frames-resident --lanes $lanes --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'MP/s', d['config']['frames_per_s'], 'frames/s', d['parity_gate'])"
done
done
