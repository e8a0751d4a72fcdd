#!/usr/bin/env python3
"""One rank of an 8-GPU run, measured on ONE GPU.

The 16384^2 bench frame over eight participants = eight rows of DC groups (jxlt_shard_rect).  Participant 0 is a real
device context that holds ITS rectangle (1/8 of the frame) and runs the real exchange protocol
(jxlt_shard_encode: host/frame_shards.cc); participants 1..7 run the same protocol on slab operations (jxlt_slab_ops)
that REPLAY what a device context would have produced for their rectangles -- histograms, section sizes, section
bytes, recorded beforehand on this GPU with the product's own kernels -- and answer at once.  So participant 0 never
waits for a slower peer, and what is timed is what one rank of an 8-GPU job has on its critical path:

    its kernels (1/8 of the frame)  ->  the serial stage (histogram sum, code construction on participants 0 and 1,
    code tables back)  ->  section packing  ->  sizes to participant 0, layout back  ->  its share of the bytes over
    its PCIe link  ->  header + TOC

The codestream is checked against the single-GPU codestream of the whole frame.  Not a scaling curve (there is one GPU
here): the only hardware-backed piece of the N = 8 projection that this box can give.
Usage: slab_of_8.py [size] [frames] [out.json]"""
import ctypes as C
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__  # noqa: E402
import bench  # noqa: E402


class ReplayRecord(C.Structure):
    """tools/slab_replay.cc: slab_replay_record"""
    _fields_ = [("dc_hist", C.POINTER(C.c_uint32)), ("ac_hist", C.POINTER(C.c_uint32)),
                ("bytes", C.POINTER(C.c_uint8) * 2), ("off", C.POINTER(C.c_uint64) * 2), ("bits", C.POINTER(C.c_uint32) * 2),
                ("nsec", C.c_size_t * 2)]


def replay_lib():
    """The seven replaying participants as native threads (tools/slab_replay.cc), built on demand."""
    here = ROOT / "tools"
    so, src = here / "libslab_replay.so", here / "slab_replay.cc"
    if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-I" + str(ROOT / "include"), "-o", str(so), str(src),
                        "-L" + str(ROOT / "libjxl-tiny_amd" / "host"), "-ljxltiny_host",
                        "-L" + str(ROOT / "libjxl-tiny_amd" / "csrc"), "-ljxltiny_hip", "-pthread",
                        "-Wl,-rpath," + str(ROOT / "libjxl-tiny_amd" / "host"), "-Wl,-rpath," + str(ROOT / "libjxl-tiny_amd" / "csrc")],
                       check=True)
    lib = C.CDLL(str(so))
    lib.slab_replay_start.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(ReplayRecord),
                                      C.c_size_t, C.c_size_t, C.c_float, C.c_int]
    lib.slab_replay_join.restype = C.c_int
    return lib


def main():
    import numpy as np
    import torch
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    world, d = 8, 1.0
    pkg = __graft_entry__.load_package()
    dev = torch.device("cuda", 0)
    frame = bench.frame_rows_on_device(torch, size, 0, size, 0, dev)
    torch.cuda.synchronize()

    # ---- the whole frame on one GPU: the bytes to reproduce, and the time to compare with
    enc = pkg.Encoder(0)
    enc.set_device_image([frame[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=frame)
    for _ in range(5):
        single = enc.encode_resident(d, copy=False)
    t0 = time.perf_counter()
    for _ in range(10):
        single = enc.encode_resident(d, copy=False)
    one_gpu_ms = (time.perf_counter() - t0) / 10 * 1e3
    single = single.tobytes()
    enc.close()

    # ---- record what a device context produces for every rectangle (the product's kernels, this GPU)
    rects = [pkg.shard_rect(size, size, world, r) for r in range(world)]
    encs, hists = [], []
    for (x0, y0, x1, y1) in rects:
        e = pkg.Encoder(0)
        e.set_device_image([frame[c, y0:, x0:].data_ptr() for c in range(3)], size * 4, x1 - x0, y1 - y0, keepalive=frame)
        e.enqueue(d)
        hists.append(e.fetch_histograms())
        encs.append(e)
    ac_sum = sum(h[0].astype(np.uint64) for h in hists).astype(np.uint32)
    dc_sum = sum(h[1].astype(np.uint64) for h in hists).astype(np.uint32)
    ac_table, dc_table = pkg.build_code_tables(ac_sum, dc_sum)
    records = []
    for r, e in enumerate(encs):
        rec = {"ac_hist": np.ascontiguousarray(hists[r][0].reshape(-1)), "dc_hist": np.ascontiguousarray(hists[r][1].reshape(-1)),
               "sections": [e.pack_sections(0, dc_table), e.pack_sections(1, ac_table)]}
        records.append(rec)
    for e in encs[1:]:
        e.close()
    enc0 = encs[0]  # participant 0 keeps its context (and its rectangle)

    # ---- the protocol: participant 0 on the device (this thread), 1..7 replaying as native threads
    name = "/jxlt-slab8-%d" % os.getpid()
    sections = ((size + 2047) // 2048) ** 2 + ((size + 255) // 256) ** 2
    capacity = max(32 << 20, size * size // 4)
    group0 = pkg.ShardGroup(name, 0, world, capacity, sections + 64)
    total = frames + 5
    recs = (ReplayRecord * (world - 1))()
    keep = []
    for r in range(1, world):
        rec, out = records[r], recs[r - 1]
        out.dc_hist = rec["dc_hist"].ctypes.data_as(C.POINTER(C.c_uint32))
        out.ac_hist = rec["ac_hist"].ctypes.data_as(C.POINTER(C.c_uint32))
        for kind in (0, 1):
            data, off, bits = rec["sections"][kind]
            data, off, bits = np.ascontiguousarray(data), np.ascontiguousarray(off, np.uint64), np.ascontiguousarray(bits, np.uint32)
            keep.append((data, off, bits))
            out.bytes[kind] = data.ctypes.data_as(C.POINTER(C.c_uint8))
            out.off[kind] = off.ctypes.data_as(C.POINTER(C.c_uint64))
            out.bits[kind] = bits.ctypes.data_as(C.POINTER(C.c_uint32))
            out.nsec[kind] = len(bits)
    lib = replay_lib()
    if lib.slab_replay_start(name.encode(), world, 1, capacity, sections + 64, recs, size, size, C.c_float(d), total) != 0:
        raise SystemExit("slab_replay_start failed")
    step_ms, kernel_ms, stages, same = [], [], [], True
    for i in range(total):
        t0 = time.perf_counter()
        view = group0.encode(enc0, size, size, d)
        dt = (time.perf_counter() - t0) * 1e3
        if i >= 5:
            step_ms.append(round(dt, 3))
            kernel_ms.append({k: round(v, 3) for k, v in enc0.kernel_times().items()})
            stages.append(group0.last_timeline())
        if i in (0, total - 1):
            same = same and view.tobytes() == single
    errors = ["%d replaying participant(s) failed" % n for n in [lib.slab_replay_join()] if n]
    group0.close()
    enc0.close()
    srt = sorted(step_ms)
    med = lambda v: sorted(v)[len(v) // 2]
    doc = {"what": "participant 0 of 8 (its 1/8 rectangle of the %dx%d frame on the GPU, the real jxlt_shard_encode protocol) with "
                   "participants 1..7 replaying recorded device results at once: one rank's critical path of an 8-GPU frame" % (size, size),
           "frames": frames, "rect_of_participant_0": list(rects[0]),
           "ms_per_frame": round(sum(step_ms) / len(step_ms), 3), "ms_per_frame_median": srt[len(srt) // 2], "ms_per_frame_min": srt[0],
           "kernel_ms_median": {k: med([km[k] for km in kernel_ms]) for k in kernel_ms[0]},
           "whole_frame_on_this_gpu_ms": round(one_gpu_ms, 3),
           "ratio_whole_frame_over_one_rank_median": round(one_gpu_ms / srt[len(srt) // 2], 2),
           "step_ms": step_ms, "slowest_step_over_median": round(srt[-1] / srt[len(srt) // 2], 2),
           "host_stage_ms_median": {k: round(med([st[k] for st in stages]), 3) for k in stages[0]} if stages and stages[0] else None,
           "stage_split_ms_median": (lambda m: {"kernels_until_ac_histogram": round(m["ac_histogram"], 3),
                                                "sums_and_codes": round(m["code_tables"] - m["ac_histogram"], 3),
                                                "own_sizes": round(m["own_sizes"] - m["code_tables"], 3),
                                                "layout": round(m["layout"] - m["own_sizes"], 3),
                                                "hand_over_and_placed": round(m["all_placed"] - m["layout"], 3)})(
               {k: med([st[k] for st in stages]) for k in stages[0]}) if stages and stages[0] else None,
           "same_bytes_as_single_gpu": bool(same), "errors": errors,
           "note": "serial stage + hand-over = ms_per_frame_median - kernels; the seven replaying participants are native threads "
                   "(tools/slab_replay.cc; round 5: Python threads taking turns on the interpreter lock); the peers answer at once, so nothing here is waiting for a slower "
                   "GPU; every participant's section bytes land in one shared output buffer (participant 0's by DMA from the GPU, the "
                   "others' by memcpy)"}
    line = json.dumps(doc)
    print(line)
    if out_path:
        Path(out_path).write_text(json.dumps(doc, indent=1) + "\n")


if __name__ == "__main__":
    main()
