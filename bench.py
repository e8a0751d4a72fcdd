#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the MI355X-native JPEG XL tiny encoder.

    python bench.py --gpus N --steps K --warmup W [--size S]

One "step" = one full encode of ONE synthetic S x S linear-sRGB frame (default 16384 x 16384,
BASELINE config #4) that is already resident in HBM as planar f32, ending with the complete
.jxl codestream bytes in host memory.

N = 1   jxlt_encode_resident_view on one GPU:
          device   tile_kernel (XYB, adaptive quant, chroma-from-luma, strategy search, quantise),
                   DC-group tokenisation, group_scan + token_kernel (AC tokens + histograms)
          host     DC / AC histograms -> prefix codes
          device   pack_tile_measure (exact section sizes) -> pack_tile_write, copied in ranges to
                   the page-locked output buffer while the host writes header + TOC in front
N > 1   the SAME frame, cut into row slabs of whole DC groups (2048 rows), slab r resident on GPU r
        (BASELINE config #4: "groups sharded by index across the GPUs, host-side assembly").  One
        process per GPU; the ranks meet in a POSIX shared-memory segment (jxlt_shard_group_*): the
        2 x 64 x 64 histograms are summed by rank 0, which hands the code tables back; every GPU
        packs its sections and copies them straight into its byte range of the one output buffer.
        No RCCL on the data path (torch.distributed/RCCL only for the barrier and the max-reduce
        of the timing).  scaling = "strong" (the frame is fixed, N grows).
        --replicas: every rank encodes its own S x S frame instead ("weak").

Launch: for N > 1 either through `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`
or bare (`python bench.py --gpus N`): without WORLD_SIZE in the environment the script starts that
launcher itself as a child process, before anything touches a GPU, and relays its output.

PFM file I/O and the H2D upload are outside the timed region of `value`; the PCIe-inclusive rate of the
metric as written (PFM payload in page-locked host memory -> .jxl bytes) is reported beside it as
`pfm_inclusive` (N = 1).  Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

# CPU hygiene of the harness (before numpy / torch are imported).  The pool's boxes show 256 CPUs and grant a process
# 16 at a time (cgroup cpu.max): the OpenMP / MKL thread pools that numpy and torch size by the CPU count keep
# spinning for a while after their last parallel region, the control group runs out of its quota and EVERY thread of
# the process -- the encoding thread too -- is stopped for the rest of the 100 ms period.  Seen as steps of 12-18 ms
# among steps of 5.2 (tools/outlier_probe.sh: two throttlings per run, in the warm-up or the first timed steps).
for _k, _v in (("OMP_NUM_THREADS", "8"), ("MKL_NUM_THREADS", "8"), ("OPENBLAS_NUM_THREADS", "8"),
               ("OMP_WAIT_POLICY", "PASSIVE"), ("KMP_BLOCKTIME", "0"), ("GOMP_SPINCOUNT", "0")):
    os.environ.setdefault(_k, _v)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PCIE_PEAK_GBS = 63.0   # same guide: PCIe Gen5 x16 host link
ALGO_BYTES_PER_PIXEL = 12.0  # 3 planes x f32, each pixel read once
CHUNK_ROWS = 1024  # generator granularity: the frame's content does not depend on how it is cut into slabs


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=16384, help="frame is size x size pixels")
    ap.add_argument("--distance", type=float, default=1.0)
    ap.add_argument("--cpu-sample", type=int, default=8192,
                    help="edge of the crop of the frame the CPU oracle encodes (cpu_baseline)")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--replicas", action="store_true",
                    help="N > 1: one independent frame per rank (weak scaling) instead of one frame over all ranks")
    ap.add_argument("--no-extras", action="store_true", help="skip cpu_baseline / pfm_inclusive / parity legs")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="N > 1: frames in flight over the group (jxlt_shard_pipeline_*): the timed steps then measure "
                         "THROUGHPUT with this many frames overlapping, not the time of one frame (the default, 1, "
                         "is the headline: one config-#4 frame at a time; its JSON line reports the two-in-flight "
                         "rate beside it as in_flight_2)")
    ap.add_argument("--frame-batch", type=int, default=0,
                    help="secondary workload (BASELINE config #5: --frame-batch 256 --gpus 8; PCIe-inclusive, never the "
                         "headline value): a step is a batch of this many --frame-size frames in page-locked HOST "
                         "memory, frame i on GPU i mod N, every GPU's share through jxlt_batch_encoder_run (uploads, "
                         "kernels and downloads of different frames overlap)")
    ap.add_argument("--frame-size", default="3840x2160")
    ap.add_argument("--lanes", type=int, default=3, help="device contexts per GPU of the frame-batch encoder")
    ap.add_argument("--frames-resident", action="store_true",
                    help="frame-batch workload with the frames already in HBM (no PCIe upload)")
    return ap.parse_args()


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as children (nothing in this process has
    touched a GPU yet, and it never will)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def frame_rows_on_device(torch, size, y0, y1, seed, device):
    """Rows [y0, y1) of the SURVEY.md 8(d) synthetic frame, evaluated on the GPU (float64 math, N(0, 0.02) noise
    from a torch generator seeded per 1024-row chunk, so that any row range gives the same samples as the whole
    frame).  Returns a [3, y1 - y0, size] float32 tensor."""
    assert y0 % CHUNK_ROWS == 0
    out = torch.empty((3, y1 - y0, size), dtype=torch.float32, device=device)
    x = torch.arange(size, dtype=torch.float64, device=device)[None, :]
    gen = torch.Generator(device=device)
    for c0 in range(y0, y1, CHUNK_ROWS):
        c1 = min(y1, c0 + CHUNK_ROWS)
        gen.manual_seed(1234 + 1000003 * seed + c0 // CHUNK_ROWS)
        y = torch.arange(c0, c1, dtype=torch.float64, device=device)[:, None]
        r = 0.5 + 0.4 * torch.sin(x / 37) * torch.cos(y / 53)
        g = 0.5 + 0.4 * torch.sin((x + y) / 91)
        b = 0.3 + 0.3 * torch.cos(x / 19 - y / 29)
        m = 0.6 + 0.4 * ((torch.floor(x / 48) + torch.floor(y / 80)) % 2)
        for c, p in enumerate((r, g, b)):
            v = p * m + torch.randn(p.shape, dtype=torch.float64, device=device, generator=gen) * 0.02
            out[c, c0 - y0:c1 - y0] = (v.clamp_(0, 1) ** 2.2).to(torch.float32)
    return out


def make_frame_on_device(torch, size, seed, device):
    """The whole size x size frame (tools/ use this)."""
    return frame_rows_on_device(torch, size, 0, size, seed, device)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(relaunch_under_torchrun(args))

    import numpy as np
    import torch
    torch.set_num_threads(8)  # (see the note on CPU hygiene at the top)
    import __graft_entry__
    pkg = __graft_entry__.load_package()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    # (test hook: JXLT_BENCH_ONE_DEVICE=1 runs every rank on cuda:0 with a gloo process group, so that the
    # N > 1 control flow can be exercised on a single-GPU box; not a measurement mode)
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

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if one_device else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.frame_batch > 0:
        run_frame_batch(args, np, torch, pkg, dist, barrier, max_over_ranks, rank, world, dev_index, device)
        return

    size, d = args.size, args.distance
    sharded = world > 1 and not args.replicas
    if sharded:
        # this rank's rectangle of whole DC groups (16384^2 over 1 / 2 / 4 / 8: rows of DC groups)
        x0, y0, x1, y1 = pkg.shard_rect(size, size, world, rank)
    else:
        x0, y0, x1, y1 = 0, 0, size, size
    seed = rank if (world > 1 and args.replicas) else 0
    # (experiment knob BENCH_CONTEXT_FIRST=1, tools/alloc_order_probe.sh: the device context made BEFORE the frame is
    # generated -- the library's allocations are then the first of the process, and tile_kernel takes 4.75 instead of
    # 4.05 ms: where the buffers land in device memory decides that much)
    enc = pkg.Encoder(dev_index) if os.environ.get("BENCH_CONTEXT_FIRST") else None
    slab = frame_rows_on_device(torch, size, y0, y1, seed, device) if y1 > y0 else None
    if slab is not None and (x0, x1) != (0, size):
        slab = slab[:, :, x0:x1].contiguous()  # (the rectangle alone stays resident)
    torch.cuda.synchronize()
    if enc is None:
        enc = pkg.Encoder(dev_index)
    if slab is not None:
        enc.set_device_image([slab[c].data_ptr() for c in range(3)], (x1 - x0) * 4, x1 - x0, y1 - y0, keepalive=slab)

    group = None
    if sharded:
        # the ranks' meeting point; rank 0 creates it, the others attach after the barrier
        name = "/jxlt-bench-%s" % os.environ.get("MASTER_PORT", "0")
        sections = ((size + 2047) // 2048) ** 2 + ((size + 255) // 256) ** 2
        # output area in /dev/shm: a quarter byte per pixel (the 16384^2 bench frame needs 0.09), within what
        # the shared-memory file system has free
        capacity = max(32 << 20, size * size // 4)
        try:
            st = os.statvfs("/dev/shm")
            capacity = min(capacity, max(16 << 20, st.f_bavail * st.f_frsize // 2))
        except OSError:
            pass
        if rank == 0:
            group = pkg.ShardGroup(name, 0, world, capacity, sections + 64)
        barrier()
        if rank != 0:
            group = pkg.ShardGroup(name, rank, world, capacity, sections + 64)

    def step():
        if group is not None:
            return group.encode(enc, size, size, d)
        # the codestream is assembled in the context's page-locked host buffer (no extra copy)
        return enc.encode_resident(d, num_threads=args.host_threads, copy=False)

    def open_pipeline(depth, tag):
        """jxlt_shard_pipeline_*: rank 0 creates the lanes' segments, the others attach behind the barrier."""
        pname = "%s-%s" % (name, tag)
        pipe = pkg.ShardPipeline(pname, 0, world, dev_index, depth, capacity, sections + 64) if rank == 0 else None
        barrier()
        if pipe is None:
            pipe = pkg.ShardPipeline(pname, rank, world, dev_index, depth, capacity, sections + 64)
        return pipe

    def run_pipelined(pipe, frames, check=None):
        """`frames` encodes of the frame with pipe.depth of them in flight; returns the seconds they took (max over
        ranks, barriers on both sides).  check(bytes): called on rank 0 with every codestream."""
        ptrs = [slab[c].data_ptr() for c in range(3)] if slab is not None else None
        rows = (y1 - y0) if slab is not None else 0
        barrier()
        t = time.perf_counter()
        tickets = []
        for k in range(frames):
            if k >= pipe.depth:
                v = pipe.wait(tickets[k - pipe.depth])
                if check is not None and v is not None:
                    check(v)
            tickets.append(pipe.submit_device(ptrs, (x1 - x0) * 4, size, size, rows, d))
        for tk in tickets[max(0, frames - pipe.depth):]:
            v = pipe.wait(tk)
            if check is not None and v is not None:
                check(v)
        barrier()
        return max_over_ranks(time.perf_counter() - t), v

    # The encoding thread -- this one -- next to its GPU for the timed region (the application's choice: the library
    # binds only threads it owns; pinned near / far / not at all: 5.185 / 5.22 / 5.24 ms, profiles/r04_driver_cmd_box2_*);
    # the mask is put back before the CPU-side legs, which use every core.
    affinity_before = pkg.bind_thread_near_device(dev_index) if os.environ.get("JXLT_NO_AFFINITY") is None else None

    # N = 1 with the extra legs: the device-only leg (the kernels without code construction and packing, K launches) runs
    # HERE, in front of the warm-up -- the GPU's clock needs some 75 ms under load to reach its level, which five warm-up
    # steps of 5 ms are not (BENCH_r04 and the round-5 runs of the driver's command: tile_kernel 4.0 -> 3.75 ms over the
    # first ten steps), and of the line's legs this is the one that is only kernels.  
    # The line says so itself (`harness`: what ran in front, for how long, the first and the last timed step's
    # tile_kernel time -- VERDICT r5 item 6 / ADVICE r5: a reader of the JSON alone can tell clock ramp from code).
    result_early = {}
    pre_warm_ms = 0.0
    if not sharded and not args.no_extras and slab is not None:
        t_pre = time.perf_counter()
        device_only_leg(args, enc, result_early, size * size / 1e6)
        pre_warm_ms = 1e3 * (time.perf_counter() - t_pre)
    pipelined = sharded and args.in_flight > 1
    ktimes = {}
    step_ms, warmup_ms, step_kernel_ms, step_copy, step_stages = [], [], [], [], []
    probe = None
    if pipelined:
        pipe = open_pipeline(args.in_flight, "timed")
        run_pipelined(pipe, max(args.warmup, args.in_flight))
        elapsed, jxl = run_pipelined(pipe, args.steps)
        if jxl is not None:
            jxl = jxl.tobytes()
        pipe.close()
    else:
        tw = time.perf_counter()
        for _ in range(args.warmup):
            jxl = step()
            tn = time.perf_counter()
            warmup_ms.append(round(1e3 * (tn - tw), 3))
            tw = tn
        barrier()
        # Per-stage device times of the timed steps themselves: HIP events that the C ABI records on the encoder's
        # own stream around every stage of every encode (read after each step).  The time of every single step
        # (this rank's clock) and the stage times of the first eight go into the line as well: a run whose early
        # steps are slow shows whether the kernels were (clocks still rising) or the host's share was.
        # ... and what could make a step slow that is not the library's doing: the control group's CPU
        # throttling (cpu.stat, read through a descriptor opened beforehand), the encoding thread's context switches
        # (getrusage of the thread: no file), and the host time inside the longest copy command of the step
        # (jxlt_encode_stats: a call that meets the runtime creating a copy engine's queue takes milliseconds).  A few
        # microseconds per step, all of them read between two steps.
        probe = StepProbe()
        probe.sample()
        # (step_ms = the encode call alone; the diagnostic reads behind it -- HIP-event times of the step that has just
        # ended, counters, the host's stage times -- are inside `elapsed`, i.e. counted against `value`, and their
        # cost is reported as harness.diagnostic_calls_ms_per_step)
        diag_s = 0.0
        t0 = time.perf_counter()
        tp = t0
        for i in range(args.steps):
            jxl = step()
            tn = time.perf_counter()
            step_ms.append(round(1e3 * (tn - tp), 3))
            if slab is not None:
                kt = enc.kernel_times()
                for k, v in kt.items():
                    ktimes[k] = ktimes.get(k, 0.0) + v / args.steps
                step_kernel_ms.append({k: round(v, 3) for k, v in kt.items()})
                st = enc.stats()
                step_copy.append((st["copy_calls"], round(st["longest_copy_call_us"], 1)))
                tl = pkg.last_frame_timeline()
                if tl is not None:
                    step_stages.append({k: round(v, 3) for k, v in tl.items()})
            probe.sample()
            tp = time.perf_counter()
            diag_s += tp - tn
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        if jxl is not None:
            jxl = jxl.tobytes()

    if affinity_before is not None:
        try:
            os.sched_setaffinity(0, affinity_before)
        except OSError:
            pass
    frames = world if (world > 1 and args.replicas) else 1
    mpix = size * size / 1e6
    value = frames * mpix * args.steps / elapsed
    tile_ms = ktimes.get("tile_kernel", float("nan"))
    slab_pixels = (y1 - y0) * (x1 - x0)
    achieved = ALGO_BYTES_PER_PIXEL * slab_pixels / (tile_ms * 1e-3) / 1e9 if slab_pixels else float("nan")

    # ---- legs every rank takes part in (outside the timed region)
    per_rank_kernel_ms = None
    in_flight_2 = None
    if dist is not None:
        per_rank_kernel_ms = [None] * world
        dist.all_gather_object(per_rank_kernel_ms, {k: round(v, 3) for k, v in ktimes.items()})
    if sharded and not args.no_extras and not pipelined:
        # two frames in flight over the same ranks (jxlt_shard_pipeline_*): what the serial stage of a sharded frame
        # costs in throughput when the next frame's kernels run beside it
        mismatches = []
        pipe = open_pipeline(2, "if2")
        nfr = max(6, args.steps)
        run_pipelined(pipe, 4)
        secs, _ = run_pipelined(pipe, nfr, check=(lambda v: mismatches.append(1) if v.tobytes() != jxl else None))
        pipe.close()
        in_flight_2 = {"frames": nfr, "ms_per_frame": round(1e3 * secs / nfr, 3),
                       "value": round(size * size / 1e6 * nfr / secs, 2), "unit": "Mpixels/s",
                       "speedup_over_one_at_a_time": round((elapsed / args.steps) / (secs / nfr), 3),
                       "same_bytes": not mismatches,
                       "note": "throughput with the kernels of frame k + 1 running beside the code construction and "
                               "packing of frame k; one frame at a time stays the headline"}

    if rank != 0:
        if dist is not None:
            dist.barrier()  # rank 0's legs (single-GPU cross-check, CPU baseline)
            dist.destroy_process_group()
        if group is not None:
            group.close()
        return

    jxl_bytes = jxl
    result = {
        "metric": "Mpixels/s encode (PFM->.jxl), frame resident in HBM, codestream bytes in host memory",
        "value": round(value, 2),
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True,
        "scaling": "strong" if sharded else "weak",
        "in_flight": args.in_flight if sharded else 1,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": ("ONE %dx%d synthetic linear-sRGB frame%s, distance %.2f, full 8x8/16x8/8x16 strategy search + "
                         "adaptive quant + chroma-from-luma" %
                         (size, size, " sharded over %d GPUs in rectangles of whole DC groups (here: rows of DC groups; BASELINE config #4)" % world
                          if sharded else (" per GPU (independent replicas)" if world > 1 else ""), d)),
            "parallelism": ("DC groups -> GPUs by index (rectangles of whole DC groups); host-side histogram sum on rank 0 (shared memory), every "
                            "GPU writes its sections into one output buffer; no RCCL on the data path" if sharded else
                            "one independent frame per rank, no data-path collective" if world > 1 else "single GPU"),
            "rect_on_rank0": [x0, y0, x1, y1], "codestream_bytes": len(jxl_bytes),
            "codestream_sha256": hashlib.sha256(jxl_bytes).hexdigest()[:16]},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     # (the committed PMC profile is of the whole frame on one GPU; a slab's launch moves its share of
                     # those bytes: the kernel's traffic is per tile, tiles do not share data beyond the halo columns)
                     "traffic": (None if pmc_traffic(size) is None else
                                 int(pmc_traffic(size) * slab_pixels / float(size * size))),
                     "kernel": "tile12_kernel", "kernel_ms": round(tile_ms, 3),
                     "algorithmic_bytes_per_launch": ALGO_BYTES_PER_PIXEL * slab_pixels,
                     "note": "rank 0's launch (its slab of the frame)" if sharded else "whole frame"},
        "kernel_ms": {k: round(v, 3) for k, v in ktimes.items()},
    }
    if step_ms:
        tk = [k.get("tile_kernel") for k in step_kernel_ms]
        result["harness"] = {
            "pre_warm": ("device-only leg (kernels without code construction and packing, %d launches) in FRONT of the "
                         "warm-up steps: most of the GPU's clock ramp lies in front of the timed region; what is left of it shows as "
                         "tile_kernel_ms_first_minus_last" % (3 + max(3, args.steps))
                         if pre_warm_ms else "none (warm-up steps only)"),
            "pre_warm_ms": round(pre_warm_ms, 1),
            "warmup_steps_ms_total": round(sum(warmup_ms), 1),
            "tile_kernel_ms_first_timed_step": tk[0] if tk else None,
            "tile_kernel_ms_last_timed_step": tk[-1] if tk else None,
            "tile_kernel_ms_first_minus_last": round(tk[0] - tk[-1], 3) if tk and tk[0] is not None and tk[-1] is not None else None,
            "diagnostic_calls_ms_per_step": round(1e3 * diag_s / max(1, len(step_ms)), 4),
            "note": "ms_per_step and value include the diagnostic calls; step_ms lists the encode calls alone"}
        srt = sorted(step_ms)
        result["ms_per_step_median"] = srt[len(srt) // 2] if len(srt) % 2 else round((srt[len(srt) // 2 - 1] + srt[len(srt) // 2]) / 2, 3)
        result["ms_per_step_min"] = srt[0]
        result["step_ms"] = step_ms
        result["warmup_step_ms"] = warmup_ms
        result["kernel_ms_per_step"] = step_kernel_ms
        result["step_diagnostics"] = step_diagnostics(step_ms, step_kernel_ms, step_copy, probe, step_stages)
    # (traffic and instruction counts come from committed counter profiles, not from this run: say which, and
    # whether the device code has changed since they were collected)
    _, fresh = pmc_traffic(size, with_doc=True)
    if fresh is not None:
        result["roofline"]["traffic_profile"] = fresh
    if per_rank_kernel_ms is not None:
        result["kernel_ms_per_rank"] = per_rank_kernel_ms
    if in_flight_2 is not None:
        result["in_flight_2"] = in_flight_2
    if pipelined:
        result["config"]["workload"] += "; THROUGHPUT with %d frames in flight (not one frame's time)" % args.in_flight
    valu = pmc_valu(size)
    if world == 1 and valu is not None and tile_ms == tile_ms:
        # Supplementary: the kernel is VALU-issue bound, not HBM bound.  Instructions per wave from the
        # committed counter profile, duration measured live; peak = one wave64 VALU instruction per 2 cycles and SIMD
        # (MI355X_MICROARCH.md "Wave scheduling", confirmed by tools/valu_issue_probe.hip -> profiles/): 256 CUs x 4 SIMDs
        # x 2.4 GHz / 2.
        peak = 256 * 4 * 2.4e9 / 2 / 1e12
        ach = valu["valu_insts_per_wave"] * valu["waves"] / (tile_ms * 1e-3) / 1e12
        result["roofline"]["valu_issue"] = {"achieved": round(ach, 4), "peak": round(peak, 4),
                                            "unit": "T wave64 VALU instructions/s", "frac": round(ach / peak, 4),
                                            "valu_insts_per_wave": valu["valu_insts_per_wave"],
                                            "profile": valu["freshness"]}

    if not args.no_extras:
        if sharded:
            # the sharded codestream must be the single-GPU codestream of the same frame, byte for byte
            full = frame_rows_on_device(torch, size, 0, size, 0, device)
            enc1 = pkg.Encoder(dev_index)
            enc1.set_device_image([full[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=full)
            single = enc1.encode_resident(d, copy=False).tobytes()
            result["parity_gate"] = {"sharded_equals_single_gpu_codestream": single == jxl_bytes,
                                     "single_gpu_sha256": hashlib.sha256(single).hexdigest()[:16]}
            if in_flight_2 is not None and not in_flight_2["same_bytes"]:
                result["parity_gate"]["sharded_equals_single_gpu_codestream"] = False
            enc1.close()
            # (no cpu_baseline at N > 1: it is a figure of the N = 1 line, and the other ranks would sit in the barrier
            # below for the ten seconds it takes)
            del full
        else:
            result.update(result_early)
            extras_single_gpu(args, np, torch, pkg, enc, slab, dev_index, device, result)
    print(json.dumps(result), flush=True)
    # (the ranks are released and the shared-memory segment closed whatever the gate says: a rank 0 that left
    # here would leave the others in their barrier until the launcher kills them)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if group is not None:
        group.close()
    gate = result.get("parity_gate", {})
    if gate.get("groups_mismatching", 0) or gate.get("sharded_equals_single_gpu_codestream") is False:
        raise SystemExit("parity gate failed: %s" % json.dumps(gate))


class StepProbe:
    """Between two timed steps: the control group's throttling counters and the calling thread's context switches.
    Raw samples are kept; deltas per step are worked out behind the timed region."""

    def __init__(self):
        import resource
        self._resource = resource
        self._fd = None
        for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
            try:
                self._fd = os.open(path, os.O_RDONLY)
                break
            except OSError:
                pass
        self.samples = []

    def sample(self):
        ru = self._resource.getrusage(self._resource.RUSAGE_THREAD)
        raw = b""
        if self._fd is not None:
            try:
                raw = os.pread(self._fd, 512, 0)
            except OSError:
                pass
        self.samples.append((ru.ru_nvcsw, ru.ru_nivcsw, raw))

    def deltas(self):
        def parse(raw):
            out = {}
            for line in raw.decode("ascii", "replace").splitlines():
                parts = line.split()
                if len(parts) == 2 and parts[0] in ("nr_throttled", "throttled_usec", "throttled_time"):
                    out[parts[0]] = int(parts[1])
            return out
        res = []
        for a, b in zip(self.samples, self.samples[1:]):
            pa, pb = parse(a[2]), parse(b[2])
            thr_us = (pb.get("throttled_usec", 0) - pa.get("throttled_usec", 0)) or \
                     (pb.get("throttled_time", 0) - pa.get("throttled_time", 0)) // 1000
            res.append({"voluntary_ctxsw": b[0] - a[0], "involuntary_ctxsw": b[1] - a[1],
                        "nr_throttled": pb.get("nr_throttled", 0) - pa.get("nr_throttled", 0), "throttled_usec": thr_us})
        return res


def step_diagnostics(step_ms, step_kernel_ms, step_copy, probe, step_stages=()):
    """Why was a step slow?  For every timed step above 1.15 x the median: what differs from the run's typical step --
    device time (the kernels' HIP-event times of that very step: clocks, pre-emption, another tenant on the device),
    the host's view of the frame's stages (jxlt_last_frame_timeline: when the DC histogram, the AC histogram, the
    codes, the sizes and the last byte arrived -- a stage that took longer while the kernels did not is a queue
    delay in front of a kernel, a slow link, or the host), CPU throttling of the control group, involuntary context
    switches of the encoding thread, the longest copy command's host time.  `cause` names whatever explains at least a
    fifth of the excess; "unattributed" otherwise."""
    srt = sorted(step_ms)
    median = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    mean = sum(step_ms) / len(step_ms)
    deltas = probe.deltas() if probe is not None else []

    def med(vals):
        v = sorted(vals)
        return v[len(v) // 2] if v else 0.0
    kernel_sum = [sum(k.values()) for k in step_kernel_ms]
    kmed = med(kernel_sum)
    copy_med = med([c[1] for c in step_copy])
    # stage DURATIONS from the timeline's time stamps: device pipeline up to the DC histogram, tokenisation behind it,
    # code construction, section sizes, hand-over of the bytes
    names = ("until_dc_histogram", "dc_to_ac_histogram", "codes", "sizes", "hand_over")
    keys = ("dc_histogram_ms", "ac_histogram_ms", "codes_ms", "sizes_ms", "done_ms")
    durs = []
    for tl in step_stages:
        t = [tl[k] for k in keys]
        durs.append({names[0]: t[0], names[1]: max(0.0, t[1] - t[0]), names[2]: max(0.0, t[2] - t[1]),
                     names[3]: max(0.0, t[3] - t[2]), names[4]: max(0.0, t[4] - t[3])})
    dur_med = {n: med([d[n] for d in durs]) for n in names} if durs else {}
    out = {"median_ms": round(median, 3), "mean_minus_median_ms": round(mean - median, 3),
           "host_stage_ms_median": {n: round(v, 3) for n, v in dur_med.items()},
           "threshold": "steps above 1.15 x the median", "slow_steps": [],
           "totals": {"nr_throttled": sum(d["nr_throttled"] for d in deltas),
                      "throttled_usec": sum(d["throttled_usec"] for d in deltas),
                      "involuntary_ctxsw": sum(d["involuntary_ctxsw"] for d in deltas),
                      "voluntary_ctxsw": sum(d["voluntary_ctxsw"] for d in deltas),
                      "longest_copy_call_us": max([c[1] for c in step_copy], default=0.0)}}
    for i, ms in enumerate(step_ms):
        if ms <= 1.15 * median:
            continue
        excess = ms - median
        entry = {"step": i, "ms": ms, "excess_ms": round(excess, 3)}
        cause = {}
        if i < len(kernel_sum):
            entry["kernel_ms"] = step_kernel_ms[i]
            dk = kernel_sum[i] - kmed
            if dk > 0.2 * excess:
                cause["device_kernels_ms"] = round(dk, 3)
        if i < len(durs):
            entry["host_stage_ms"] = {n: round(durs[i][n], 3) for n in names}
            for n in names:
                dd = durs[i][n] - dur_med[n]
                if dd > 0.2 * excess:
                    cause["host_stage_%s_ms_over_median" % n] = round(dd, 3)
        if i < len(step_copy):
            entry["copy_calls"], entry["longest_copy_call_us"] = step_copy[i]
            if (step_copy[i][1] - copy_med) * 1e-3 > 0.2 * excess:
                cause["copy_call_on_host_ms"] = round((step_copy[i][1] - copy_med) * 1e-3, 3)
        if i < len(deltas):
            entry.update(deltas[i])
            if deltas[i]["throttled_usec"] * 1e-3 > 0.2 * excess or deltas[i]["nr_throttled"]:
                cause["cgroup_throttled_ms"] = round(deltas[i]["throttled_usec"] * 1e-3, 3)
            if deltas[i]["involuntary_ctxsw"]:
                cause["involuntary_ctxsw"] = deltas[i]["involuntary_ctxsw"]
        if i > 0 and step_ms[i - 1] > 1.15 * median and "device_kernels_ms" in cause:
            cause["follows_a_slow_step"] = True  # (the GPU's clock coming back up)
        entry["cause"] = cause if cause else {"unattributed_ms": round(excess, 3)}
        out["slow_steps"].append(entry)
    return out


def device_only_leg(args, enc, result, mpix):
    """The device pipeline without code construction and packing, max(3, K) launches behind three untimed ones."""
    d = args.distance
    for _ in range(3):
        enc.enqueue(d, 0)
    enc.synchronize()
    reps = max(3, args.steps)
    t1 = time.perf_counter()
    for _ in range(reps):
        enc.enqueue(d, 0)
    enc.synchronize()
    result["device_only_mpix_s"] = round(mpix / ((time.perf_counter() - t1) / reps), 1)


def extras_single_gpu(args, np, torch, pkg, enc, frame, dev_index, device, result):
    """N = 1 legs outside the timed region: device-only rate, parity gate over the whole frame, CPU baseline,
    and the metric as written (PFM payload in page-locked host memory -> .jxl bytes)."""
    import ctypes as C
    import jxlt_testlib as T
    size, d = args.size, args.distance
    mpix = size * size / 1e6

    # ---- device-only rate (the pipeline without code construction and packing): measured in front of the warm-up
    # (device_only_leg); the raw tokens the parity gate reads come from one more pass
    enc.enqueue(d, 0)
    enc.synchronize()
    fr = enc.fetch_raw()
    result["config"]["raw_token_bytes"] = int(fr.group_token_offset[fr.num_groups])

    # ---- parity gate: groups sampled over the WHOLE frame (corners, last row / column, an interior lattice)
    # against the oracle run on the matching 256 x 256 crops (an AC group depends on nothing outside itself,
    # SURVEY.md F9): token bytes, quantised DC, quant field, strategies, chroma-from-luma factors
    gpg = (size + 255) // 256
    lattice = sorted(set([0, gpg - 1] + [int(round(i * (gpg - 1) / 7.0)) for i in range(8)]))
    picks = sorted(set((gy, gx) for gy in lattice for gx in lattice))
    offs = np.ctypeslib.as_array(fr.group_token_offset, shape=(fr.num_groups + 1,)).copy()
    xb = fr.xsize_blocks

    def grid(ptr, dtype, pitch, rows):
        return np.ctypeslib.as_array(ptr, shape=(rows * pitch,)).view(dtype).reshape(rows, pitch)

    qdc = [grid(fr.quant_dc[c], np.int16, xb, fr.ysize_blocks) for c in range(3)]
    rq = grid(fr.raw_quant_field, np.uint8, xb, fr.ysize_blocks)
    st = grid(fr.ac_strategy, np.uint8, xb, fr.ysize_blocks)
    ytox = grid(fr.ytox_map, np.int8, fr.xsize_tiles, fr.ysize_tiles)
    ytob = grid(fr.ytob_map, np.int8, fr.xsize_tiles, fr.ysize_tiles)
    bad = 0
    t2 = time.perf_counter()
    for gy, gx in picks:
        crop = np.ascontiguousarray(frame[:, gy * 256:(gy + 1) * 256, gx * 256:(gx + 1) * 256].cpu().numpy())
        want = T.oracle_hot_path(crop, d)
        g = gy * gpg + gx
        got = C.string_at(C.addressof(fr.tokens.contents) + int(offs[g]), int(offs[g + 1] - offs[g]))
        hb, wb = want.raw_quant.shape
        by, bx = gy * 32, gx * 32
        ok = (got == want.group_tokens[0] and
              all(np.array_equal(qdc[c][by:by + hb, bx:bx + wb], want.quant_dc[c]) for c in range(3)) and
              np.array_equal(rq[by:by + hb, bx:bx + wb], want.raw_quant) and
              np.array_equal(st[by:by + hb, bx:bx + wb], want.strategy) and
              np.array_equal(ytox[gy * 4:gy * 4 + want.ytox.shape[0], gx * 4:gx * 4 + want.ytox.shape[1]], want.ytox) and
              np.array_equal(ytob[gy * 4:gy * 4 + want.ytob.shape[0], gx * 4:gx * 4 + want.ytob.shape[1]], want.ytob))
        bad += not ok
    result["parity_gate"] = {"groups_checked": len(picks), "groups_mismatching": int(bad),
                             "sample": "%d groups on an %dx%d lattice over the whole frame incl. corners and last "
                                       "row/column; tokens + side-band grids vs the oracle on the same crops, %.1f s"
                                       % (len(picks), len(lattice), len(lattice), time.perf_counter() - t2)}

    cpu_baseline_leg(args, np, pkg, T, frame, dev_index, result)

    # ---- the metric as written: PFM payload (interleaved, bottom-up f32) in page-locked HOST memory -> .jxl bytes;
    # the upload is part of the encode, pipelined in DC-group rows under tile_kernel (jxlt_image_attach_host_pfm)
    try:
        payload, owner = pkg.pinned_empty((size * size * 3,), np.float32)
    except pkg.JxlTinyError:
        payload = None
    if payload is not None:
        rows = 2048
        view = payload.reshape(size, size, 3)
        for r0 in range(0, size, rows):  # bottom-up: image row y is payload row size - 1 - y
            blk = frame[:, r0:r0 + rows].permute(1, 2, 0).flip(0).contiguous().cpu().numpy()
            view[size - r0 - blk.shape[0]:size - r0] = blk
        enc2 = pkg.Encoder(dev_index)
        for _ in range(2):
            enc2.attach_host_pfm(payload, size, size)
            pfm_jxl = enc2.encode_resident(d, num_threads=args.host_threads, copy=False)
        reps2 = 3
        t5 = time.perf_counter()
        for _ in range(reps2):
            enc2.attach_host_pfm(payload, size, size)
            pfm_jxl = enc2.encode_resident(d, num_threads=args.host_threads, copy=False)
        pfm_s = (time.perf_counter() - t5) / reps2
        same = hashlib.sha256(pfm_jxl.tobytes()).hexdigest()[:16] == result["config"]["codestream_sha256"]
        enc2.close()
        del view, payload, owner
        bound = PCIE_PEAK_GBS / ALGO_BYTES_PER_PIXEL * 1e3
        result["pfm_inclusive"] = {
            "workload": "%dx%d PFM payload in page-locked host memory -> row-wise upload under the kernels -> .jxl "
                        "bytes in host memory" % (size, size),
            "ms_per_frame": round(pfm_s * 1e3, 2), "value": round(mpix / pfm_s, 1), "unit": "Mpixels/s",
            "h2d_gb_s": round(ALGO_BYTES_PER_PIXEL * size * size / pfm_s / 1e9, 1),
            "pcie_bound_mpix_s": round(bound, 1), "frac_of_pcie_bound": round(mpix / pfm_s / bound, 3),
            "same_bytes_as_resident_encode": bool(same)}
        if not same:
            result["parity_gate"]["groups_mismatching"] += 1


def cpu_baseline_leg(args, np, pkg, T, frame, dev_index, result):
    """cpu_baseline: the oracle (pixel pipeline + bitstream stage) on a bounded crop of the frame on this host's
    cores, one thread and all cores; the GPU's codestream of that crop must be the oracle's bytes."""
    size, d = args.size, args.distance
    result.setdefault("parity_gate", {}).setdefault("groups_mismatching", 0)
    # a bounded, group-aligned crop of the same frame (centre of the frame)
    s = min(args.cpu_sample, size)
    s -= s % 256 if s >= 256 else 0
    o = ((size - s) // 2) // 256 * 256
    crop = np.ascontiguousarray(frame[:, o:o + s, o:o + s].cpu().numpy())
    cpu_jxl, pix_s, bs_s = T.oracle_encode_file(crop, d, nthreads=1)
    cpu_s = pix_s + bs_s
    result["cpu_baseline"] = {
        "value": round(s * s / 1e6 / cpu_s, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port",
        "sample": "centre %dx%d crop of the benchmark frame: oracle pixel pipeline (%.2f s) + oracle bitstream stage "
                  "(%.2f s), both plain C, one thread -- the reference's own structure" % (s, s, pix_s, bs_s),
        "cpu": _cpu_model(), "host_cores": os.cpu_count(), "codestream_bytes": len(cpu_jxl)}
    # the same crop with the reference's independent units -- the 256 x 256 groups -- spread over ALL host cores by
    # POSIX threads inside the oracle (orc_encode_hot_path_threads); the bitstream stage stays serial like the
    # reference's OptimizeSections.  Same work, same bytes.
    ng = s // 256
    if ng >= 2:
        # (the CPUs this process may run on, and how many of them its control group lets it use at a time: on the
        # GPU boxes of this pool 16 of 256 -- threads beyond the quota are throttled, 256 threads ran at a third of
        # the rate of 16, tools/cpu_scaling_probe.py)
        try:
            usable = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            usable = os.cpu_count() or 1
        quota = _cgroup_cpu_quota()
        nthr = max(1, min(ng * ng, usable if quota is None else min(usable, quota)))
        par_jxl, ppix_s, pbs_s = T.oracle_encode_file(crop, d, nthreads=nthr)
        par_s = ppix_s + pbs_s
        result["cpu_baseline"]["all_cores"] = {
            "value": round(s * s / 1e6 / par_s, 1), "unit": "Mpixels/s", "cores": nthr,
            "sample": "same crop and same work: %d groups over %d POSIX threads (%.3f s) + the serial bitstream stage "
                      "(%.3f s)" % (ng * ng, nthr, ppix_s, pbs_s),
            "pixel_pipeline_only_mpix_s": round(s * s / 1e6 / ppix_s, 1),
            "cpus_in_affinity_mask": usable, "cgroup_cpu_quota": quota,
            "same_bytes_as_one_thread": par_jxl == cpu_jxl}
        # the GPU's codestream of that crop must be those bytes too
        gpu_crop = pkg.encode_file(crop, d, device=dev_index)
        result["parity_gate"]["crop_codestream_equals_oracle"] = gpu_crop == cpu_jxl
        if gpu_crop != cpu_jxl:
            result["parity_gate"]["groups_mismatching"] += 1



def run_frame_batch(args, np, torch, pkg, dist, barrier, max_over_ranks, rank, world, dev_index, device):
    """Secondary workload (BASELINE config #5): batches of frames from page-locked host memory (PCIe-inclusive),
    `--frame-batch` frames per GPU and step; one batch encoder (several lanes) per rank / GPU."""
    import jxlt_testlib as T
    w, h = (int(v) for v in args.frame_size.lower().split("x"))
    distinct = max(1, min(args.frame_batch // max(world, 1), 8))
    frames, owners = [], []
    hh = (h + CHUNK_ROWS - 1) // CHUNK_ROWS * CHUNK_ROWS
    for i in range(distinct):
        full = frame_rows_on_device(torch, max(w, 16), 0, hh, 100 * rank + i + 1, device)
        if args.frames_resident:
            frames.append(full[:, :h, :w].contiguous())
        else:
            arr, owner = pkg.pinned_empty((3, h, w))
            arr[...] = full[:, :h, :w].cpu().numpy()
            frames.append(arr)
            owners.append(owner)
        del full
    enc = pkg.BatchEncoder(dev_index, lanes=args.lanes)
    total = args.frame_batch
    mine = list(range(rank, total, world))  # frames round-robin over the GPUs
    n = len(mine)
    descs, keep = enc.describe([frames[i % distinct] for i in mine])
    first = enc.run_described(descs, n, args.distance)
    for _ in range(max(0, args.warmup - 1)):
        enc.run_described(descs, n, args.distance, take=False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total_bytes = enc.run_described(descs, n, args.distance, take=False)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    mpix = total * w * h / 1e6
    h2d_gbs = 12.0 * total * w * h * args.steps / elapsed / 1e9
    result = {
        "metric": ("Mpixels/s encode, frames resident in HBM -> codestream bytes in host memory" if args.frames_resident else
                   "Mpixels/s encode, frames in page-locked host memory -> codestream bytes in host memory (PCIe-inclusive)"),
        "value": round(mpix * args.steps / elapsed, 2), "unit": "Mpixels/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "batch of %d frames %dx%d per step over %d GPU(s) (%d distinct per GPU), distance %.2f, "
                               "%d lanes per GPU" % (total, w, h, world, distinct, args.distance, args.lanes),
                   "frames_per_s": round(total * args.steps / elapsed, 1),
                   "parallelism": "independent frames, frame i on GPU i mod N, one frame queue per GPU, no collective",
                   "codestream_bytes_per_batch_rank0": int(total_bytes)},
        "roofline": {"bound": "pcie", "achieved": round(h2d_gbs / world, 2), "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                     "frac": round(h2d_gbs / world / PCIE_PEAK_GBS, 4), "traffic": None,
                     "note": "host->device bytes of the frames (12 B/pixel) per GPU; peak = PCIe 5.0 x16 payload rate"},
    }
    if rank == 0:
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


def kernel_source_sha16():
    """Fingerprint of the device code (every file under libjxl-tiny_amd/csrc but built objects): the counter
    profiles under profiles/ store the fingerprint of the sources they were collected with, and the bench line
    says whether that still is what runs (there is no .git on the GPU box to ask for a commit)."""
    h = hashlib.sha256()
    for p in sorted((ROOT / "libjxl-tiny_amd" / "csrc").iterdir()):
        if p.suffix in (".h", ".hip"):
            h.update(p.name.encode())
            h.update(p.read_bytes())
    return h.hexdigest()[:16]


def profile_freshness(path, doc):
    """Which committed counter profile a figure comes from and whether it still describes what runs: `stale_reason` is
    null only when the profile carries the fingerprint of the CURRENT device sources; otherwise it says why not, and
    stderr says it too (the check fails loudly)."""
    stored = doc.get("kernel_source_sha16")
    now = kernel_source_sha16()
    reason = None
    if stored is None:
        reason = "the profile carries no fingerprint of the device sources it was collected with"
    elif stored != now:
        reason = ("collected with device sources %s, the sources now are %s: the kernels have changed since -- "
                  "run tools/collect_profiles.sh" % (stored, now))
    if reason is not None:
        print("bench.py: WARNING: profiles/%s is stale: %s" % (Path(path).name, reason), file=sys.stderr, flush=True)
    return {"file": "profiles/" + Path(path).name, "collected_with_kernel_source_sha16": stored,
            "current_kernel_source_sha16": now, "stale": reason is not None, "stale_reason": reason}


def pmc_traffic(size, with_doc=False):
    """HBM bytes per tile_kernel launch from the committed PMC profile of this workload
    (profiles/*_traffic_<size>.json, made by tools/collect_traffic.py from separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes with the calibrated gfx950 correction);
    None when no profile of this frame size is committed."""
    best = None
    used = None
    for p in sorted((ROOT / "profiles").glob("*_traffic_%d.json" % size)):
        try:
            doc = json.load(open(p))
            best = doc["kernels"]["tile_kernel"]["hbm_bytes"]
            used = (p, doc)
        except (OSError, KeyError, ValueError):
            pass
    # (the freshness of the profile that is USED -- the latest round's --, not of every older one beside it)
    fresh = profile_freshness(*used) if used is not None and with_doc else None
    return (best, fresh) if with_doc else best


def pmc_valu(size):
    """VALU instructions per wave of tile_kernel from the committed counter profile
    (profiles/*_tile_valu_<size>.json, made with tools/tile_cycles.sh); None if absent."""
    best, path = None, None
    for p in sorted((ROOT / "profiles").glob("*_tile_valu_%d.json" % size)):
        try:
            doc = json.load(open(p))
            doc["valu_insts_per_wave"], doc["waves"]
            best, path = doc, p
        except (OSError, KeyError, ValueError):
            pass
    if best is not None:
        best["freshness"] = profile_freshness(path, best)
    return best


def _cgroup_cpu_quota():
    """CPUs the process's control group may use at a time (cgroup v2 cpu.max / v1 cfs quota), rounded up; None if
    unlimited or unknown."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1, -(-int(quota) // int(period)))
        return None
    except (OSError, ValueError):
        pass
    try:
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return max(1, -(-quota // period)) if quota > 0 and period > 0 else None
    except (OSError, ValueError):
        return None


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
