for cfg in "" "JXLT_PACK_TWO_PASS=0" "JXLT_PACK_TWO_PASS=1"; do
echo -n "noise 8192 [$cfg] "; env $cfg timeout 300 python tools/run_resident.py 8192 30 1.0 noise 2>&1 | grep done | cut -c1-60
echo -n "8192 d0.5 [$cfg] "; env $cfg timeout 300 python tools/run_resident.py 8192 30 0.5 2>&1 | grep done | cut -c1-60
echo -n "8192 d1 [$cfg] "; env $cfg timeout 300 python tools/run_resident.py 8192 30 1.0 2>&1 | grep done | cut -c1-60
echo -n "4096 d0.1 [$cfg] "; env $cfg timeout 300 python tools/run_resident.py 4096 30 0.1 2>&1 | grep done | cut -c1-60
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
