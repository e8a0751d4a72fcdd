import sys, time
sys.path.insert(0, '.')
import torch
import __graft_entry__, bench
pkg = __graft_entry__.load_package()
size = 16384
f = bench.make_frame_on_device(torch, size, 0, torch.device("cuda", 0))
torch.cuda.synchronize()
def row(d, new_enc=True, enc=None):
    if enc is None:
        enc = pkg.Encoder(0)
        enc.set_device_image([f[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=f)
    for _ in range(6):
        enc.encode_resident(d, copy=False)
    t0 = time.perf_counter()
    for _ in range(10):
        enc.encode_resident(d, copy=False)
    dt = (time.perf_counter() - t0) / 10
    print("d=%g %.3f ms" % (d, dt * 1e3), enc.kernel_times(), flush=True)
    return enc
import os
if os.environ.get("BIND"):
    L = pkg.hip_lib()
    L.jxlt_bind_thread_near_device.argtypes = [__import__("ctypes").c_int]
    print("bind rc", L.jxlt_bind_thread_near_device(0), "cpus", sorted(os.sched_getaffinity(0))[:4], len(os.sched_getaffinity(0)))
order = [float(x) for x in sys.argv[1:]]
for d in order:
    e = row(d)
    e.close()
if os.environ.get("VARIANT") == "keep":
    print("-- second context while the first is alive")
    a = row(0.5)
    b = row(0.5)
    print("-- the first again")
    row(0.5, enc=a)
if os.environ.get("VARIANT") == "reuse":
    print("-- one context, distances in turn")
    a = row(0.5)
    row(1.0, enc=a)
    row(0.5, enc=a)
if os.environ.get("VARIANT") == "grow":
    print("-- one context: 8192^2 frames first, then 16384^2 (its buffers are freed and allocated again, larger)")
    enc = pkg.Encoder(0)
    half = f[:, :8192, :8192].contiguous()
    enc.set_device_image([half[c].data_ptr() for c in range(3)], 8192 * 4, 8192, 8192, keepalive=half)
    for _ in range(6):
        enc.encode_resident(0.5, copy=False)
    enc.set_device_image([f[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=f)
    row(0.5, enc=enc)
