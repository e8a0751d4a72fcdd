timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for sz in 1024 2048 4096 16384; do echo -n "$sz: "; timeout 300 python tools/run_resident.py $sz 80 2>&1 | grep done | cut -c1-50; done; done
