for rep in 1 2; do
for d in 1.0 4.0; do
for cfg in "" "JXLT_PACK_TWO_PASS=1"; do
echo -n "16384 d$d [$cfg] "; env $cfg timeout 300 python tools/run_resident.py 16384 30 $d 2>&1 | grep done | cut -c1-60
done; done; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
