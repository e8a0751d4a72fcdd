import sys, time
sys.path.insert(0, '.')
import __graft_entry__ as G
import torch
torch.zeros(1, device='cuda'); torch.cuda.synchronize()
pkg = G.load_package()
for i in range(3):
    t0 = time.perf_counter(); e = pkg.Encoder(0); t1 = time.perf_counter()
    print("context %d created in %.1f ms" % (i, (t1 - t0) * 1e3)); e.close()
