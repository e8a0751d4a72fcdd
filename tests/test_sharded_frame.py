"""One frame sharded over two processes (world_size 2, gloo, CPU only): the orchestration of
libjxl-tiny_amd/sharded.py -- histogram all_reduce, identical code tables, per-rank section
packing, gather, assembly on rank 0 -- must give byte-for-byte the single-process codestream.

Here the per-slab "device" is replaced by test infrastructure (oracle tokens, oracle DC
tokeniser, reference bit packer) so that the test runs without a GPU; the GPU variant
(test_gpu_parity.py::test_sharded_slabs_on_one_gpu) uses real device contexts."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

import jxlt_testlib as T

W, H, D = 72, 2048 + 2048 + 40, 1.0  # three DC groups tall -> slabs of 2 + 1 DC groups


class OracleSlab:
    """Slab 'encoder' built from the oracle + reference packer (tests only)."""

    def __init__(self, planes, distance):
        self.res = T.oracle_hot_path(planes, distance)
        self.dc_records = T.oracle_dc_records(self.res)

    def histograms(self):
        ac = T.token_histogram(self.res.all_tokens())
        dc = sum((T.token_histogram(r) for r in self.dc_records))
        return ac, dc

    def pack(self, ac_table, dc_table):
        def pk(sections, table):
            packed = T.pack_sections_python(sections, table)
            data = np.frombuffer(b"".join(p[0] for p in packed), np.uint8)
            off = np.zeros(len(packed) + 1, np.uint64)
            off[1:] = np.cumsum([len(p[0]) for p in packed])
            return data, off, np.array([p[1] for p in packed], np.uint32)
        return pk(self.dc_records, dc_table), pk(self.res.group_tokens, ac_table)


def _worker(rank, world, port, q):
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pkg = T.product()
        import importlib.util
        spec = importlib.util.spec_from_file_location("jxlt_sharded", str(T.PKG / "sharded.py"))
        sharded = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(sharded)
        planes = T.to_planes(T.synthetic_image(W, H))
        y0, y1 = sharded.slab_rows(H, world)[rank]
        slab = OracleSlab(np.ascontiguousarray(planes[:, y0:y1]), D)
        out = sharded.encode_sharded(slab, sharded.TorchComm(dist), W, H, D, pkg)
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
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert results[1] is None
    assert not isinstance(results[0], str), results[0]
    planes = T.to_planes(T.synthetic_image(W, H))
    want = T.assemble_codestream(T.oracle_hot_path(planes, D), D)
    assert results[0] == want


def test_slab_rows_cover_whole_dc_groups():
    import importlib.util
    spec = importlib.util.spec_from_file_location("jxlt_sharded", str(T.PKG / "sharded.py"))
    sharded = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharded)
    for h, world in [(16384, 8), (16384, 3), (5000, 2), (100, 4), (2049, 2)]:
        rows = sharded.slab_rows(h, world)
        assert rows[0][0] == 0 and rows[-1][1] == h
        for (a0, a1), (b0, b1) in zip(rows, rows[1:]):
            assert a1 == b0 and (a1 % 2048 == 0 or a1 == h)
