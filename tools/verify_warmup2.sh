for i in 1 2 3; do timeout 600 python3 tools/config_table.py 2>/dev/null | python3 -c "
import json,sys
print([json.loads(l)['ms_per_frame'] for l in sys.stdin])"; done
CFGS="JXLT_COPY_WARMUP=3" bash tools/outlier_probe.sh 2>&1 | head -8
