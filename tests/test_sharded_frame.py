"""One frame sharded over several participants -- the product's C-level protocol
(libjxl-tiny_amd/host/frame_shards.cc: jxlt_shard_group_* / jxlt_shard_encode_ops) on CPU.

The protocol (histogram sum on participant 0, code tables handed back, every participant's
sections placed in its byte range of ONE shared output buffer, header + TOC in front) is the
same code the GPU ranks of bench.py run; only the per-slab "device" is replaced through the
jxlt_slab_ops hook by test infrastructure (oracle tokens, oracle DC tokeniser, reference bit
packer), so the test needs no GPU.  Expected bytes: the oracle's whole-frame codestream.

* world_size 2, two processes, torch.distributed/gloo for rendezvous + barrier (the launch
  shape of `torchrun bench.py --gpus 2`);
* 3 and 8 participants as threads of one process attached to one segment (uneven slabs,
  participants without rows)."""
import ctypes as C
import multiprocessing as mp
import os
import socket
import threading

import numpy as np
import pytest

import jxlt_testlib as T

W, H, D = 72, 2048 + 2048 + 40, 1.0  # three DC-group rows


class OracleSlab:
    """Slab operations (jxlt_slab_ops) built from the oracle + the reference packer (tests only)."""

    def __init__(self, pkg, planes, distance):
        self.pkg, self.planes, self.distance = pkg, planes, distance
        fn = pkg._SLAB_FN
        self.ops = pkg.SlabOps()
        self._cb = {k: fn[k](getattr(self, "_" + k)) for k in fn}  # keep the thunks alive
        for k, cb in self._cb.items():
            setattr(self.ops, k, cb)

    def _enqueue(self, _self, params):
        assert abs(params.contents.distance - self.distance) < 1e-7
        self.res = T.oracle_hot_path(self.planes, self.distance)
        self.dc_records = T.oracle_dc_records(self.res)
        return 0

    def _dc_histogram(self, _self, out):
        self.dc_hist = np.ascontiguousarray(sum((T.token_histogram(r) for r in self.dc_records)), np.uint32)
        out[0] = self.dc_hist.ctypes.data_as(C.POINTER(C.c_uint32))
        return 0

    def _begin_dc_pack(self, _self, table):
        self.dc_table = np.ctypeslib.as_array(table, shape=(4096,)).copy()
        return 0

    def _ac_histogram(self, _self, out):
        self.ac_hist = np.ascontiguousarray(T.token_histogram(self.res.all_tokens()), np.uint32)
        out[0] = self.ac_hist.ctypes.data_as(C.POINTER(C.c_uint32))
        return 0

    def _measure(self, _self, table, dc, ac):
        ac_table = np.ctypeslib.as_array(table, shape=(4096,)).copy()
        self.packed, self.keep = [], []
        for sections, tab, dst in ((self.dc_records, self.dc_table, dc), (self.res.group_tokens, ac_table, ac)):
            packed = T.pack_sections_python(sections, tab)
            off = np.zeros(len(packed) + 1, np.uint64)
            off[1:] = np.cumsum([len(p[0]) for p in packed])
            bits = np.array([p[1] for p in packed], np.uint32)
            self.packed.append(b"".join(p[0] for p in packed))
            self.keep += [off, bits]
            dst.contents.bytes = None
            dst.contents.section_offset = off.ctypes.data_as(C.POINTER(C.c_uint64))
            dst.contents.section_bits = bits.ctypes.data_as(C.POINTER(C.c_uint32))
            dst.contents.num_sections = len(packed)
        return 0

    def _write(self, _self, out, dc_runs, n_dc, ac_runs, n_ac):
        base = C.addressof(out.contents)
        for kind, runs, n in ((0, dc_runs, n_dc), (1, ac_runs, n_ac)):
            off = self.keep[2 * kind]
            for i in range(n):
                r = runs[i]
                lo, hi = int(off[r.first_section]), int(off[r.first_section + r.num_sections])
                C.memmove(base + r.dst_offset, self.packed[kind][lo:hi], hi - lo)
        return 0

    def _finish(self, _self):
        return 0


def _test_frame(w, h, flat=False):
    """The survey's generator, or (flat) a grey frame with a lattice of dots: few tokens, so that the reference bit
    packer -- Python, a record at a time -- stays cheap on frames of several DC groups."""
    if not flat:
        return T.to_planes(T.synthetic_image(w, h))
    img = np.full((h, w, 3), 0.25, np.float32)
    img[::61, ::67] = (0.4, 0.3, 0.2)
    return T.to_planes(img)


def _participant(pkg, name, rank, world, w, h, d, after_create=None, flat=False):
    planes = _test_frame(w, h, flat)
    x0, y0, x1, y1 = pkg.shard_rect(w, h, world, rank)
    grp = pkg.ShardGroup(name, rank, world, 4 << 20, 4096) if rank == 0 else None
    if after_create:
        after_create()
    if grp is None:
        grp = pkg.ShardGroup(name, rank, world, 4 << 20, 4096)
    slab = OracleSlab(pkg, np.ascontiguousarray(planes[:, y0:y1, x0:x1]), d)
    out = []
    for _ in range(2):  # two frames through the same group: the control block is reusable
        view = grp.encode_ops(slab.ops, w, h, d)
        out.append(view.tobytes() if view is not None else None)
        # the stage times the protocol keeps per rank (testing header; tools/slab_of_8.py): in protocol order
        tl = grp.last_timeline()
        stamps = [tl[k] for k in grp.STAGES[:7 if rank else 8]]
        assert all(b >= a >= 0.0 for a, b in zip(stamps, stamps[1:])), tl
    if after_create:
        after_create()  # rank 0 must not unlink the segment while others still use it
    grp.close()
    return out


def _worker(rank, world, port, name, q, w=W, h=H):
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        out = _participant(T.product(), name, rank, world, w, h, D, after_create=dist.barrier)
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put((rank, "ERROR: %r" % (e,)))


def test_two_process_sharded_frame_equals_single_process(built):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "/jxlt-test-%d" % os.getpid()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert not isinstance(results[0], str), results[0]
    assert not isinstance(results[1], str), results[1]
    assert results[1] == [None, None]
    planes = T.to_planes(T.synthetic_image(W, H))
    want = T.assemble_codestream(T.oracle_hot_path(planes, D), D)
    assert results[0] == [want, want]


def test_two_process_frame_of_one_row_of_dc_groups(built):
    """World size 2 over gloo, a frame that is ONE row of DC groups (2 x 1): until round 3 such a frame could not be
    sharded at all (row slabs only); now every rank takes a DC group, and rank 1's sections are runs in the middle of
    the codestream (DC groups, then its AC groups row by row between rank 0's)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "/jxlt-test-x-%d" % os.getpid()
    w, h = 2048 + 300, 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q, w, h)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert not isinstance(results[0], str), results[0]
    assert not isinstance(results[1], str), results[1]
    planes = T.to_planes(T.synthetic_image(w, h))
    want = T.assemble_codestream(T.oracle_hot_path(planes, D), D)
    assert results[0] == [want, want] and results[1] == [None, None]


@pytest.mark.parametrize("world,w,h", [(3, 40, 2048 + 2048 + 40), (8, 40, 2048 + 300), (2, 40, 2048 * 2),
                                       (8, 7 * 2048 + 100, 24),          # ONE row of 8 DC groups, one each
                                       (8, 3 * 2048 + 9, 2048 + 70),     # 4 x 2 DC groups over 8: two bands of four
                                       (5, 2 * 2048 + 300, 2048 + 20),   # 3 x 2 over 5: the bands are cut differently
                                       (3, 2048 + 40, 16)])              # 2 x 1 DC groups over 3: one participant idles
def test_threads_attached_to_one_segment(built, world, w, h):
    """Uneven slabs (3 participants, 3 DC-group rows of which the last is short), participants
    without work (8 participants, 2 rows), exact multiples; frames that are cut along x as well (round 4: DC groups
    shard by index -- a one-row frame over 8 participants, 4 x 4 DC groups over 8, 3 x 3 over 5): a participant's
    sections are then several runs of the codestream."""
    d = 2.0
    flat = w > 2048 and h > 2048
    name = "/jxlt-test-thr-%d-%d" % (os.getpid(), world)
    barrier = threading.Barrier(world)
    results = [None] * world

    def run(rank):
        try:
            results[rank] = _participant(built, name, rank, world, w, h, d, after_create=barrier.wait, flat=flat)
        except Exception as e:  # pragma: no cover
            results[rank] = "ERROR: %r" % (e,)
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    planes = _test_frame(w, h, flat)
    want = T.assemble_codestream(T.oracle_hot_path(planes, d), d)
    assert results[0] == [want, want], results[0] if isinstance(results[0], str) else "codestream differs"
    for r in range(1, world):
        assert results[r] == [None, None], results[r]


def _pipeline_worker(rank, world, port, name, q):
    """Frames in flight (jxlt_shard_pipeline_*): depth 2, five frames of two different images submitted back to
    back -- frame k + 1 is in its lane's protocol while frame k is in its own."""
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pkg = T.product()
        w, h, d, depth = 56, 2048 + 2048 + 24, 2.0, 2
        x0, y0, x1, y1 = pkg.shard_rect(w, h, world, rank)
        images = [T.to_planes(T.synthetic_image(w, h, seed=900 + i)) for i in range(2)]
        # lane l always sees image l (frame k = image k % 2 = lane k % 2): one oracle-backed slab per lane
        slabs = [OracleSlab(pkg, np.ascontiguousarray(images[l][:, y0:y1, x0:x1]), d) for l in range(depth)]
        pipe = pkg.ShardPipeline(name, rank, world, -1, depth, 1 << 20, 4096, [s.ops for s in slabs]) if rank == 0 else None
        dist.barrier()
        if pipe is None:
            pipe = pkg.ShardPipeline(name, rank, world, -1, depth, 1 << 20, 4096, [s.ops for s in slabs])
        out = []
        tickets = [pipe.submit_ops(w, h, d), pipe.submit_ops(w, h, d)]  # two frames in flight
        for k in range(2, 5):
            v = pipe.wait(tickets[k - 2])
            out.append(v.tobytes() if v is not None else None)
            tickets.append(pipe.submit_ops(w, h, d))
        for t in tickets[3:]:
            v = pipe.wait(t)
            out.append(v.tobytes() if v is not None else None)
        dist.barrier()
        pipe.close()
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put((rank, "ERROR: %r" % (e,)))


def test_two_process_pipeline_with_two_frames_in_flight(built):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "/jxlt-test-pipe-%d" % os.getpid()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert not isinstance(results[0], str), results[0]
    assert not isinstance(results[1], str), results[1]
    assert results[1] == [None] * 5
    want = [T.assemble_codestream(T.oracle_hot_path(T.to_planes(T.synthetic_image(56, 2048 + 2048 + 24, seed=900 + i)), 2.0), 2.0)
            for i in range(2)]
    assert results[0] == [want[k % 2] for k in range(5)]
    # nothing is left in /dev/shm: rank 0 drops the segments' names as soon as every rank has mapped them
    assert not [f for f in os.listdir("/dev/shm") if f.startswith(name[1:])]


def test_pipeline_failure_releases_every_rank(built):
    """A lane whose slab operation fails on one rank: the frame's wait returns an error on BOTH ranks (no hang), and
    the other lane's frame, already in flight, still completes."""
    world, w, h, d, depth = 2, 40, 4096, 1.0, 2
    name = "/jxlt-test-pipefail-%d" % os.getpid()
    barrier = threading.Barrier(world)
    outcome = [None] * world

    def run(rank):
        planes = T.to_planes(T.synthetic_image(w, h))
        x0, y0, x1, y1 = built.shard_rect(w, h, world, rank)
        slabs = [OracleSlab(built, np.ascontiguousarray(planes[:, y0:y1, x0:x1]), d) for _ in range(depth)]
        if rank == 1:  # lane 1 of rank 1 fails in the middle of the protocol
            slabs[1]._cb["ac_histogram"] = built._SLAB_FN["ac_histogram"](lambda _s, _o: -5)
            slabs[1].ops.ac_histogram = slabs[1]._cb["ac_histogram"]
        pipe = built.ShardPipeline(name, rank, world, -1, depth, 1 << 20, 4096, [s.ops for s in slabs]) if rank == 0 else None
        barrier.wait()
        if pipe is None:
            pipe = built.ShardPipeline(name, rank, world, -1, depth, 1 << 20, 4096, [s.ops for s in slabs])
        t0, t1 = pipe.submit_ops(w, h, d), pipe.submit_ops(w, h, d)
        good = pipe.wait(t0)
        try:
            pipe.wait(t1)
            bad = None
        except built.JxlTinyError as e:
            bad = str(e)
        outcome[rank] = (good.tobytes() if good is not None else None, bad)
        barrier.wait()
        pipe.close()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    want = T.assemble_codestream(T.oracle_hot_path(T.to_planes(T.synthetic_image(w, h)), d), d)
    assert outcome[0] is not None and outcome[1] is not None, outcome
    assert outcome[0][0] == want and outcome[1][0] is None
    assert outcome[0][1] is not None and outcome[1][1] is not None, outcome


def test_shard_rectangles_tile_the_frame_in_whole_dc_groups(built):
    """jxlt_shard_rect: the participants' rectangles are disjoint, cover the frame, consist of whole DC groups, and
    the largest is as small as bands of rows cut into column ranges allow (BASELINE config #4 and the frames VERDICT
    r3 names: a one-row frame over 8, 8192^2 over 8)."""
    def dc(v):
        return (v + 2047) // 2048
    for w, h, world, largest in [(16384, 16384, 8, 8), (16384, 2048, 8, 1), (8192, 8192, 8, 2), (16384, 16384, 3, 24),
                                 (16384, 16384, 5, 15), (16384, 16384, 64, 1), (5000, 5000, 2, 6), (100, 100, 4, 1),
                                 (2049, 50, 2, 1), (2048, 2049, 3, 1), (6000, 2048 * 11, 7, 6)]:
        rects = [built.shard_rect(w, h, world, r) for r in range(world)]
        cover = np.zeros((dc(h), dc(w)), np.int32)
        areas = []
        for x0, y0, x1, y1 in rects:
            if x1 <= x0 or y1 <= y0:
                assert (x0, y0, x1, y1) == (0, 0, 0, 0)
                continue
            assert x0 % 2048 == 0 and y0 % 2048 == 0 and (x1 % 2048 == 0 or x1 == w) and (y1 % 2048 == 0 or y1 == h)
            cover[y0 // 2048:dc(y1), x0 // 2048:dc(x1)] += 1
            areas.append((dc(y1) - y0 // 2048) * (dc(x1) - x0 // 2048))
        assert (cover == 1).all(), (w, h, world)
        assert max(areas) == largest, (w, h, world, areas)
        assert len(areas) == min(world, dc(w) * dc(h))
    # rows of DC groups where that is as good (fewer runs per participant): 16384^2 over 8 = rounds 1-3's slabs
    assert [built.shard_rect(16384, 16384, 8, r) for r in range(8)] == [(0, 2048 * r, 16384, 2048 * (r + 1)) for r in range(8)]
    assert built.shard_rect(1000, 1000, 4, 0) == (0, 0, 1000, 1000)  # a single DC group belongs to participant 0


def test_failures_propagate_instead_of_hanging(built):
    """A participant whose slab operation fails must release the others with an error."""
    world, w, h, d = 2, 40, 4096, 1.0
    name = "/jxlt-test-fail-%d" % os.getpid()
    barrier = threading.Barrier(world)
    errors = [None] * world

    def run(rank):
        planes = T.to_planes(T.synthetic_image(w, h))
        x0, y0, x1, y1 = built.shard_rect(w, h, world, rank)
        grp = built.ShardGroup(name, rank, world, 1 << 20, 4096) if rank == 0 else None
        barrier.wait()
        if grp is None:
            grp = built.ShardGroup(name, rank, world, 1 << 20, 4096)
        slab = OracleSlab(built, np.ascontiguousarray(planes[:, y0:y1, x0:x1]), d)
        if rank == 1:
            slab._cb["ac_histogram"] = built._SLAB_FN["ac_histogram"](lambda _s, _o: -5)
            slab.ops.ac_histogram = slab._cb["ac_histogram"]
        try:
            grp.encode_ops(slab.ops, w, h, d)
        except built.JxlTinyError as e:
            errors[rank] = str(e)
        barrier.wait()
        grp.close()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert errors[0] is not None and errors[1] is not None, errors
