"""One frame sharded over several processes (one per GPU), SURVEY.md 8(e).

Every rank owns a row slab of whole DC groups (height a multiple of 2048 rows, the last
slab may be shorter) and runs the complete device pipeline on it; groups never exchange
pixels.  The only cross-rank dependency is the pair of global prefix codes:

    local histograms --all_reduce(SUM)--> identical code tables on every rank
    local section packing (device)  --gather--> rank 0: frame header, TOC, concatenation

Communication goes through a tiny interface (`all_reduce_sum`, `gather_arrays`) so the same
orchestration runs over torch.distributed with RCCL (GPU tensors) or gloo (CPU tests).
"""
import numpy as np


def slab_rows(frame_h, world):
    """Row ranges [(y0, y1)] per rank: whole 2048-row DC groups, balanced."""
    ndc = (frame_h + 2047) // 2048
    out, start = [], 0
    for r in range(world):
        n = ndc // world + (1 if r < ndc % world else 0)
        y0, y1 = min(frame_h, start * 2048), min(frame_h, (start + n) * 2048)
        out.append((y0, y1))
        start += n
    return out


class TorchComm:
    """all_reduce / gather over a torch.distributed process group (nccl or gloo)."""

    def __init__(self, dist, device="cpu"):
        import torch
        self.torch, self.dist, self.device = torch, dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def all_reduce_sum(self, arr):
        t = self.torch.from_numpy(arr.astype(np.int64)).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy().astype(arr.dtype)

    def gather_arrays(self, arr):
        """Variable-length uint8/uint32/uint64 arrays -> list on rank 0 (None elsewhere)."""
        torch, dist = self.torch, self.dist
        raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        n = torch.tensor([raw.size], dtype=torch.int64, device=self.device)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(sizes, n)
        sizes = [int(s.item()) for s in sizes]
        cap = max(max(sizes), 1)
        buf = torch.zeros(cap, dtype=torch.uint8, device=self.device)
        buf[:raw.size] = torch.from_numpy(raw.copy()).to(self.device)
        outs = [torch.zeros(cap, dtype=torch.uint8, device=self.device) for _ in range(self.world)]
        dist.all_gather(outs, buf)
        if self.rank != 0:
            return None
        return [o.cpu().numpy()[:s].view(arr.dtype) for o, s in zip(outs, sizes)]


def encode_sharded(slab, comm, frame_w, frame_h, distance, pkg):
    """slab: object with histograms() -> (ac, dc) uint32[64,64] and
    pack(ac_table, dc_table) -> ((dc_bytes, dc_off, dc_bits), (ac_bytes, ac_off, ac_bits))
    for this rank's rows.  Returns the codestream bytes on rank 0, None elsewhere."""
    ac_h, dc_h = slab.histograms()
    ac_h = comm.all_reduce_sum(np.ascontiguousarray(ac_h, np.uint32))
    dc_h = comm.all_reduce_sum(np.ascontiguousarray(dc_h, np.uint32))
    ac_t, dc_t = pkg.build_code_tables(ac_h, dc_h)
    dc_sec, ac_sec = slab.pack(ac_t, dc_t)
    gathered = []
    for sec in (dc_sec, ac_sec):
        data = comm.gather_arrays(np.ascontiguousarray(sec[0], np.uint8))
        sizes = comm.gather_arrays(np.diff(np.ascontiguousarray(sec[1], np.uint64)).astype(np.uint64))
        bits = comm.gather_arrays(np.ascontiguousarray(sec[2], np.uint32))
        gathered.append((data, sizes, bits))
    if comm.rank != 0:
        return None
    merged = []
    for data, sizes, bits in gathered:
        all_sizes = np.concatenate(sizes) if sizes else np.zeros(0, np.uint64)
        off = np.zeros(len(all_sizes) + 1, np.uint64)
        off[1:] = np.cumsum(all_sizes)
        merged.append((np.concatenate(data), off, np.concatenate(bits)))
    return pkg.finish_frame(frame_w, frame_h, distance, ac_h, dc_h, merged[0], merged[1])


class GpuSlab:
    """Slab encoder on a real device context (libjxl-tiny_amd.Encoder)."""

    def __init__(self, enc, distance):
        self.enc, self.distance = enc, distance

    def histograms(self):
        self.enc.enqueue(self.distance, 0)
        return self.enc.fetch_histograms()

    def pack(self, ac_table, dc_table):
        return self.enc.pack_sections(0, dc_table), self.enc.pack_sections(1, ac_table)
