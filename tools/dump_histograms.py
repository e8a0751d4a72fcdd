import sys
sys.path.insert(0, '.')
import numpy as np, torch
import __graft_entry__, bench
pkg = __graft_entry__.load_package()
size = 16384
frame = bench.make_frame_on_device(torch, size, 0, torch.device("cuda", 0))
enc = pkg.Encoder(0)
enc.set_device_image([frame[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=frame)
out = {}
for d in (1.0, 0.5, 4.0, 0.1):
    enc.enqueue(d, 0)
    ac, dc = enc.fetch_histograms()
    out["ac_%g" % d] = ac
    out["dc_%g" % d] = dc
    print(d, int((ac != 0).sum()), int((dc != 0).sum()), int(ac.sum()), int(dc.sum()))
np.savez_compressed("gpurun_out/bench_histograms.npz", **out)
