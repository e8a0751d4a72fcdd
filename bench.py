#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the MI355X-native JPEG XL tiny encoder.

One "step" = one full encode of a synthetic linear-sRGB frame that is already
resident in HBM as three planar f32 planes, ending with the complete .jxl
codestream bytes in host memory (jxlt_encode_resident_view):
  device   tile_kernel (XYB, adaptive quant, chroma-from-luma, strategy search, quantise)
           dc_elementwise_kernel, dc_chain_kernel (DC-group tokens + DC histogram)
           group_scan_kernel, token_kernel (AC tokens + AC histogram)
  host     DC histogram D2H -> DC code (while token_kernel runs); AC histogram D2H -> AC code
  device   pack_tile_measure (exact section sizes) -> pack_tile_write (sections at their final
           bit positions), copied in ranges to the page-locked output buffer while the host
           writes frame header + TOC + global sections in front of them
PFM file I/O and the H2D upload are outside the timed region (DESIGN.md quotes
the PCIe-inclusive rate separately).

    python bench.py --gpus N --steps K --warmup W [--size S]

For N > 1 launch through torch.distributed.run (one rank per GPU); every rank
encodes its own SxS frame (frames are independent units: no data-path
collective, scaling = weak); the timed region is bracketed by a barrier +
torch.cuda.synchronize() and the reported time is the max over ranks.

Prints ONE JSON line on rank 0 (see the keys at the end of main()).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALGO_BYTES_PER_PIXEL = 12.0  # 3 planes x f32, each pixel read once


def make_frame_on_device(torch, size, seed, device):
    """SURVEY.md 8(d) generator evaluated on the GPU (float64 math, torch RNG for
    the N(0, 0.02) noise), returned as a [3, size, size] float32 tensor."""
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + seed)
    out = torch.empty((3, size, size), dtype=torch.float32, device=device)
    rows = 1024
    x = torch.arange(size, dtype=torch.float64, device=device)[None, :]
    for y0 in range(0, size, rows):
        y1 = min(size, y0 + rows)
        y = torch.arange(y0, y1, dtype=torch.float64, device=device)[:, None]
        r = 0.5 + 0.4 * torch.sin(x / 37) * torch.cos(y / 53)
        g = 0.5 + 0.4 * torch.sin((x + y) / 91)
        b = 0.3 + 0.3 * torch.cos(x / 19 - y / 29)
        m = 0.6 + 0.4 * ((torch.floor(x / 48) + torch.floor(y / 80)) % 2)
        for c, p in enumerate((r, g, b)):
            v = p * m + torch.randn(p.shape, dtype=torch.float64, device=device, generator=gen) * 0.02
            out[c, y0:y1] = (v.clamp_(0, 1) ** 2.2).to(torch.float32)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=16384, help="frame is size x size pixels")
    ap.add_argument("--distance", type=float, default=1.0)
    ap.add_argument("--cpu-sample", type=int, default=8192,
                    help="edge of the top-left crop the CPU oracle encodes (baseline + parity gate)")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--shard-frame", action="store_true",
                    help="N > 1 only: the ranks' frames are the row slabs of ONE frame of size x (size*N) pixels; "
                         "histograms are all-reduced and the packed sections gathered on rank 0 "
                         "(libjxl-tiny_amd/sharded.py).  Default: one independent frame per rank.")
    ap.add_argument("--frame-batch", type=int, default=0,
                    help="secondary workload (BASELINE config #5, PCIe-inclusive, never the headline value): a step "
                         "is a batch of this many --frame-size frames in page-locked HOST memory encoded through "
                         "jxlt_batch_encoder_run (uploads, kernels and downloads of different frames overlap)")
    ap.add_argument("--frame-size", default="3840x2160")
    ap.add_argument("--lanes", type=int, default=3, help="device contexts of the frame-batch encoder")
    ap.add_argument("--frames-resident", action="store_true",
                    help="frame-batch workload with the frames already in HBM (no PCIe upload): what the lanes buy "
                         "for frames too small to fill the GPU")
    args = ap.parse_args()

    import torch
    import __graft_entry__
    pkg = __graft_entry__.load_package()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    # (test hook: JXLT_BENCH_ONE_DEVICE=1 runs every rank on cuda:0 with a gloo process group, so
    # that the N > 1 control flow can be exercised on a single-GPU box; not a measurement mode)
    one_device = os.environ.get("JXLT_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.frame_batch > 0:
        run_frame_batch(args, torch, pkg, dist, barrier, rank, world, dev_index, device, one_device)
        return

    size = args.size
    frame = make_frame_on_device(torch, size, rank, device)
    torch.cuda.synchronize()
    enc = pkg.Encoder(dev_index)
    ptrs = [frame[c].data_ptr() for c in range(3)]
    enc.set_device_image(ptrs, size * 4, size, size, keepalive=frame)

    sharded = comm = None
    if args.shard_frame and world > 1:
        import importlib.util
        spec = importlib.util.spec_from_file_location("jxlt_sharded", str(ROOT / "libjxl-tiny_amd" / "sharded.py"))
        sharded = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(sharded)
        comm = sharded.TorchComm(dist, "cpu" if one_device else device)

    def step():
        if sharded is not None:
            out = sharded.encode_sharded(sharded.GpuSlab(enc, args.distance), comm, size, size * world,
                                         args.distance, pkg)
            return out if out is not None else b""
        # the codestream is assembled in the context's page-locked host buffer (no extra copy)
        return enc.encode_resident(args.distance, num_threads=args.host_threads, copy=False)

    for _ in range(args.warmup):
        jxl = step()
    barrier()
    # Per-stage device times of the timed steps themselves: HIP events that the C ABI records on
    # the encoder's own stream around every stage of every encode (read after each step).
    ktimes = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        jxl = step()
        if sharded is None:
            for k, v in enc.kernel_times().items():
                ktimes[k] = ktimes.get(k, 0.0) + v / args.steps
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_device else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- device-only rate (the pipeline without code construction and packing)
    reps = max(3, args.steps)
    if not ktimes:  # sharded steps: separate passes
        for i in range(reps):
            enc.enqueue(args.distance, 0)
            enc.synchronize()
            for k, v in enc.kernel_times().items():
                ktimes[k] = ktimes.get(k, 0.0) + v / reps
    t1 = time.perf_counter()
    for i in range(reps):
        enc.enqueue(args.distance, 0)
    enc.synchronize()
    device_only_s = (time.perf_counter() - t1) / reps
    fr = enc.fetch_raw()
    token_bytes = int(fr.group_token_offset[fr.num_groups])

    mpix = size * size / 1e6
    value = world * mpix * args.steps / elapsed
    tile_ms = ktimes.get("tile_kernel", float("nan"))
    achieved = ALGO_BYTES_PER_PIXEL * size * size / (tile_ms * 1e-3) / 1e9

    result = {
        "metric": "Mpixels/s encode (PFM->.jxl), frame resident in HBM, codestream bytes in host memory",
        "value": round(value, 2),
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%dx%d synthetic linear-sRGB frame per GPU, distance %.2f, full 8x8/16x8/8x16 "
                               "strategy search + adaptive quant + chroma-from-luma" % (size, size, args.distance),
                   "groups_per_gpu": int(fr.num_groups), "parallelism": ("one %dx%d frame, row slabs of whole DC groups per rank; histogram all-reduce + "
                                   "section gather" % (size, size * world)) if sharded is not None else
                   "one independent frame per rank, no data-path collective",
                   "codestream_bytes": len(jxl), "raw_token_bytes": token_bytes},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(size),
                     "kernel": "tile_kernel", "kernel_ms": round(tile_ms, 3),
                     "algorithmic_bytes_per_launch": ALGO_BYTES_PER_PIXEL * size * size},
        "kernel_ms": {k: round(v, 3) for k, v in ktimes.items()},
        "device_only_mpix_s": round(mpix / device_only_s, 1),
    }

    # ---- supplementary: PCIe-inclusive rate (SURVEY.md 8(d) "end-to-end"): a frame in page-locked HOST memory ->
    # upload + encode -> codestream bytes in host memory; never the headline value.  8192 x 8192 crop (805 MB).
    if rank == 0 and size >= 8192:
        ps = 8192
        host, owner = pkg.pinned_empty((3, ps, ps))
        host[...] = frame[:, :ps, :ps].cpu().numpy()
        enc2 = pkg.Encoder(dev_index)
        for _ in range(2):
            enc2.upload(host)
            enc2.encode_resident(args.distance, num_threads=args.host_threads, copy=False)
        t3 = time.perf_counter()
        reps2 = 3
        for _ in range(reps2):
            enc2.upload(host)
            enc2.encode_resident(args.distance, num_threads=args.host_threads, copy=False)
        pcie_s = (time.perf_counter() - t3) / reps2
        enc2.close()
        del host, owner
        result["pcie_inclusive"] = {"workload": "%dx%d crop in page-locked host memory -> upload + encode -> bytes in host "
                                                "memory" % (ps, ps), "ms_per_frame": round(pcie_s * 1e3, 2),
                                    "value": round(ps * ps / pcie_s / 1e6, 1), "unit": "Mpixels/s",
                                    "h2d_gb_s": round(12.0 * ps * ps / pcie_s / 1e9, 1)}

    valu = pmc_valu(size)
    if valu is not None and tile_ms == tile_ms:
        # Supplementary: the kernel is VALU-issue bound, not HBM bound.  Instructions per wave
        # from the committed counter profile, duration measured live; peak = one wave64 VALU instruction per
        # two cycles and SIMD (MI355X_MICROARCH.md, "Wave scheduling"): 256 CUs x 4 SIMDs x 2.4 GHz / 2.
        peak = 256 * 4 * 2.4e9 / 2 / 1e12
        ach = valu["valu_insts_per_wave"] * valu["waves"] / (tile_ms * 1e-3) / 1e12
        result["roofline"]["valu_issue"] = {"achieved": round(ach, 4), "peak": round(peak, 4),
                                            "unit": "T wave64 VALU instructions/s", "frac": round(ach / peak, 4),
                                            "valu_insts_per_wave": valu["valu_insts_per_wave"]}

    if rank == 0:
        # ---- CPU baseline + parity gate on a bounded, group-aligned crop of the same frame
        import jxlt_testlib as T
        s = min(args.cpu_sample, size)
        s -= s % 256 if s >= 256 else 0
        crop = np.ascontiguousarray(frame[:, :s, :s].cpu().numpy())
        t2 = time.perf_counter()
        want = T.oracle_hot_path(crop, args.distance)
        cpu_jxl = T.assemble_codestream(want, args.distance, num_threads=1)
        cpu_s = time.perf_counter() - t2
        result["cpu_baseline"] = {
            "value": round(s * s / 1e6 / cpu_s, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": "top-left %dx%d crop of the benchmark frame: oracle hot path + host assembly, 1 thread, "
                      "%.1f s" % (s, s, cpu_s),
            "cpu": _cpu_model(), "host_cores": os.cpu_count(), "codestream_bytes": len(cpu_jxl)}
        # the same crop on many cores: 256-row strips are independent units of the hot path (no vertical
        # context crosses a group row), one oracle call per strip on a thread pool (ctypes drops the GIL)
        from concurrent.futures import ThreadPoolExecutor
        strips = [np.ascontiguousarray(crop[:, y:y + 256, :]) for y in range(0, s, 256)]
        nthr = max(1, min(len(strips), os.cpu_count() or 1))
        t3 = time.perf_counter()
        with ThreadPoolExecutor(nthr) as ex:
            par = list(ex.map(lambda p_: T.oracle_hot_path(p_, args.distance), strips))
        par_s = time.perf_counter() - t3
        strips_ok = all(par[i].group_tokens == want.group_tokens[i * (s // 256):(i + 1) * (s // 256)]
                        for i in range(len(strips))) if s >= 256 else True
        result["cpu_baseline"]["all_cores"] = {
            "value": round(s * s / 1e6 / par_s, 1), "unit": "Mpixels/s", "cores": nthr,
            "sample": "same crop, oracle hot path only (no bitstream assembly), %d strips of 256 rows on %d "
                      "threads, %.2f s" % (len(strips), nthr, par_s), "strips_equal_whole_crop": bool(strips_ok)}
        # groups of the crop must equal the same groups of the full-frame GPU encode
        gpg = (size + 255) // 256
        offs = np.ctypeslib.as_array(fr.group_token_offset, shape=(fr.num_groups + 1,)).copy()
        import ctypes as C
        bad = 0
        for gy in range(s // 256):
            for gx in range(s // 256):
                g = gy * gpg + gx
                got = C.string_at(C.addressof(fr.tokens.contents) + int(offs[g]), int(offs[g + 1] - offs[g]))
                bad += got != want.group_tokens[gy * (s // 256) + gx]
        result["parity_gate"] = {"groups_checked": (s // 256) ** 2, "groups_mismatching": int(bad)}
        print(json.dumps(result), flush=True)
        if bad:
            raise SystemExit("parity gate failed: %d groups differ from the oracle" % bad)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_frame_batch(args, torch, pkg, dist, barrier, rank, world, dev_index, device, one_device):
    """Secondary workload: batches of frames from page-locked host memory (PCIe-inclusive)."""
    w, h = (int(v) for v in args.frame_size.lower().split("x"))
    distinct = min(args.frame_batch, 8)
    side = max(w, h)
    frames, owners = [], []
    for i in range(distinct):
        full = make_frame_on_device(torch, side, 100 * rank + i, device)
        if args.frames_resident:
            frames.append(full[:, :h, :w].contiguous())
        else:
            arr, owner = pkg.pinned_empty((3, h, w))
            arr[...] = full[:, :h, :w].cpu().numpy()
            frames.append(arr)
            owners.append(owner)
        del full
    enc = pkg.BatchEncoder(dev_index, lanes=args.lanes)
    descs, keep = enc.describe([frames[i % distinct] for i in range(args.frame_batch)])
    n = args.frame_batch
    first = enc.run_described(descs, n, args.distance)
    for _ in range(max(0, args.warmup - 1)):
        enc.run_described(descs, n, args.distance, take=False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total_bytes = enc.run_described(descs, n, args.distance, take=False)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_device else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    mpix = n * w * h / 1e6
    h2d_gbs = world * 12.0 * n * w * h * args.steps / elapsed / 1e9
    result = {
        "metric": ("Mpixels/s encode, frames resident in HBM -> codestream bytes in host memory" if args.frames_resident else
                   "Mpixels/s encode, frames in page-locked host memory -> codestream bytes in host memory (PCIe-inclusive)"),
        "value": round(world * mpix * args.steps / elapsed, 2), "unit": "Mpixels/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "batch of %d frames %dx%d per GPU per step (%d distinct), distance %.2f, %d lanes"
                               % (n, w, h, distinct, args.distance, args.lanes),
                   "frames_per_s": round(world * n * args.steps / elapsed, 1),
                   "parallelism": "independent frames, round-robin over lanes and ranks, no collective",
                   "codestream_bytes_per_batch": int(total_bytes)},
        "roofline": {"bound": "pcie", "achieved": round(h2d_gbs / world, 2), "peak": 63.0, "unit": "GB/s",
                     "frac": round(h2d_gbs / world / 63.0, 4), "traffic": None,
                     "note": "host->device bytes of the frames (12 B/pixel) per GPU; peak = PCIe 5.0 x16 payload rate"},
    }
    if rank == 0:
        import jxlt_testlib as T
        f0 = frames[0].cpu().numpy() if args.frames_resident else frames[0]
        want = T.assemble_codestream(T.oracle_hot_path(np.ascontiguousarray(f0), args.distance), args.distance)
        result["parity_gate"] = {"frames_checked": 1, "frames_mismatching": int(first[0] != want)}
        print(json.dumps(result), flush=True)
        if first[0] != want:
            raise SystemExit("parity gate failed: frame 0 differs from the oracle")
    del keep, owners
    enc.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(size):
    """HBM bytes per tile_kernel launch from the committed PMC profile of this workload
    (profiles/*_traffic_<size>.json, made by tools/collect_traffic.py from separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes with the calibrated gfx950 correction);
    None when no profile of this frame size is committed."""
    best = None
    for p in sorted((ROOT / "profiles").glob("*_traffic_%d.json" % size)):
        try:
            best = json.load(open(p))["kernels"]["tile_kernel"]["hbm_bytes"]
        except (OSError, KeyError, ValueError):
            pass
    return best


def pmc_valu(size):
    """VALU instructions per wave of tile_kernel from the committed counter profile
    (profiles/*_tile_valu_<size>.json, made with tools/tile_cycles.sh); None if absent."""
    best = None
    for p in sorted((ROOT / "profiles").glob("*_tile_valu_%d.json" % size)):
        try:
            best = json.load(open(p))
            best["valu_insts_per_wave"], best["waves"]
        except (OSError, KeyError, ValueError):
            best = None
    return best


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
