# bench.py with 4 and 8 ranks, launched the way the driver launches them (torch.distributed.run), all on GPU 0
# (JXLT_BENCH_ONE_DEVICE=1): not a measurement -- the ranks share one GPU -- but the whole N-rank path runs: rectangles,
# shared output, code construction split over ranks 0 and 1, the parity gate against the single-GPU codestream.
export JXLT_BENCH_ONE_DEVICE=1
for n in 4 8; do
  echo "== $n ranks"
  timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
    bench.py --gpus $n --steps 5 --warmup 2 2>gpurun_out/ranks_$n.err | grep "^{" | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['n_gpus'], d['value'], d['ms_per_step'], d['scaling'], d['parity_gate'], d.get('in_flight_2'), d['config'].get('parallelism'))"
  tail -3 gpurun_out/ranks_$n.err
done
