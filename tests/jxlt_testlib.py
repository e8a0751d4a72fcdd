"""Shared helpers for the tests: ctypes bindings of the oracle (oracle/liboracle.so),
of the CPU execution model of the kernels (tests/hipsim/libjxlt_sim.so) and the
synthetic image generator of SURVEY.md section 8(d)."""
import contextlib
import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "libjxl-tiny_amd"
fp = C.POINTER(C.c_float)


# --------------------------------------------------------------------------- images
def synthetic_image(w, h, seed=1234, hard=False):
    """deterministic linear-sRGB test image, float32 [h, w, 3]."""
    rng = np.random.default_rng(4321 if hard else seed)
    if hard:
        return rng.random((h, w, 3)).astype(np.float32)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    r = 0.5 + 0.4 * np.sin(x / 37) * np.cos(y / 53)
    g = 0.5 + 0.4 * np.sin((x + y) / 91)
    b = 0.3 + 0.3 * np.cos(x / 19 - y / 29)
    img = np.stack([r, g, b], axis=-1)
    img *= (0.6 + 0.4 * ((np.floor(x / 48) + np.floor(y / 80)) % 2))[..., None]
    img += rng.normal(0, 0.02, img.shape)
    img = np.clip(img, 0, 1) ** 2.2
    return img.astype(np.float32)


def to_planes(img):
    """[h, w, 3] -> contiguous [3, h, w] float32."""
    return np.ascontiguousarray(np.moveaxis(img, -1, 0), dtype=np.float32)


def write_pfm(path, img, big_endian=False):
    h, w, _ = img.shape
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n%s\n" % (w, h, b"1.0" if big_endian else b"-1.0"))
        f.write(np.ascontiguousarray(img[::-1]).astype(">f4" if big_endian else "<f4").tobytes())


# --------------------------------------------------------------------------- builds
def _run(cmd, cwd):
    subprocess.run(cmd, cwd=cwd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


def build_oracle():
    _run(["make", "-s", "-C", str(ROOT / "oracle")], ROOT)
    return ROOT / "oracle" / "liboracle.so"


def build_sim(tiny_root_table=False):
    """The kernels on the CPU execution model.  tiny_root_table: a second build whose square-root
    table has 16 entries, so that ordinary images take tile_kernel's overflow path."""
    d = ROOT / "tests" / "hipsim"
    name = "libjxlt_sim_lut16.so" if tiny_root_table else "libjxlt_sim.so"
    out = d / name
    srcs = [d / "sim_encode.cc", d / "hip" / "hip_runtime.h"] + sorted((PKG / "csrc").glob("*.h"))
    if not out.exists() or any(s.stat().st_mtime > out.stat().st_mtime for s in srcs):
        # -Bsymbolic/hidden visibility: the kernels' names also exist (as HIP launch stubs) in
        # libjxltiny_hip.so; the simulator must bind to its own definitions.
        # (_FORTIFY_SOURCE off: its longjmp check does not know about fibers with stacks of their own)
        _run(["g++", "-std=c++17", "-O1", "-U_FORTIFY_SOURCE", "-D_FORTIFY_SOURCE=0", "-ffp-contract=off", "-mfma", "-fPIC", "-shared", "-I.",
              "-fvisibility=hidden", "-Wl,-Bsymbolic"] + (["-DJXLT_SQRT_LUT_SIZE=16"] if tiny_root_table else []) +
             ["-x", "c++", "sim_encode.cc", "-o", name], d)
    return out


# --------------------------------------------------------------------------- oracle
class OrcDistanceParams(C.Structure):
    _fields_ = [("distance", C.c_float), ("global_scale", C.c_int32), ("quant_dc", C.c_int32),
                ("scale", C.c_float), ("inv_scale", C.c_float), ("scale_dc", C.c_float),
                ("x_qm_scale", C.c_uint32), ("epf_iters", C.c_uint32)]


class OrcFrame(C.Structure):
    _fields_ = [("xsize", C.c_size_t), ("ysize", C.c_size_t),
                ("xsize_blocks", C.c_size_t), ("ysize_blocks", C.c_size_t),
                ("xsize_tiles", C.c_size_t), ("ysize_tiles", C.c_size_t),
                ("xsize_groups", C.c_size_t), ("ysize_groups", C.c_size_t),
                ("quant_dc", C.POINTER(C.c_int16) * 3),
                ("raw_quant_field", C.POINTER(C.c_uint8)),
                ("ac_strategy", C.POINTER(C.c_uint8)),
                ("ytox_map", C.POINTER(C.c_int8)), ("ytob_map", C.POINTER(C.c_int8)),
                ("group_tokens", C.POINTER(C.POINTER(C.c_uint8))),
                ("group_token_bytes", C.POINTER(C.c_size_t)),
                ("xyb", fp * 3), ("quant_field", fp), ("masking", fp), ("entropy8", fp)]


def _np(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).reshape(shape).copy()


class HotPathResult:
    """Plain numpy view of one hot-path run (oracle, simulator or GPU)."""
    __slots__ = ("quant_dc", "raw_quant", "strategy", "ytox", "ytob", "group_tokens",
                 "xyb", "qf", "mask", "ent8", "xsize", "ysize", "histogram", "dc_histogram", "dc_records",
                 "exact_reruns", "unsupported")

    def all_tokens(self):
        return b"".join(self.group_tokens)


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        L = C.CDLL(str(build_oracle()))
        L.orc_encode_hot_path.argtypes = [C.POINTER(fp), C.c_size_t, C.c_size_t, C.c_size_t,
                                          C.c_float, C.c_int, C.POINTER(OrcFrame)]
        L.orc_encode_hot_path.restype = C.c_int
        L.orc_encode_hot_path_threads.argtypes = [C.POINTER(fp), C.c_size_t, C.c_size_t, C.c_size_t,
                                                  C.c_float, C.c_int, C.c_int, C.c_int, C.POINTER(OrcFrame)]
        L.orc_encode_hot_path_threads.restype = C.c_int
        L.orc_frame_free.argtypes = [C.POINTER(OrcFrame)]
        L.orc_compute_distance_params.argtypes = [C.c_float, C.POINTER(OrcDistanceParams)]
        _oracle = L
    return _oracle


def distance_params(distance):
    p = OrcDistanceParams()
    oracle().orc_compute_distance_params(C.c_float(distance), C.byref(p))
    return p


def _plane_ptrs(planes):
    assert planes.dtype == np.float32 and planes.flags.c_contiguous and planes.shape[0] == 3
    arr = (fp * 3)()
    for c in range(3):
        arr[c] = planes[c].ctypes.data_as(fp)
    return arr


def set_strategy_distance(first_call_distance):
    """Oracle and CPU model: emulate a later call of a reference process whose first call used
    this distance (enc_ac_strategy.cc:178-185 static constants); 0 = off."""
    L = oracle()
    L.orc_set_strategy_distance.argtypes = [C.c_float]
    L.orc_set_strategy_distance.restype = None
    L.orc_set_strategy_distance(C.c_float(first_call_distance))
    S = _sim_lib()
    S.sim_set_strategy_distance.argtypes = [C.c_float]
    S.sim_set_strategy_distance.restype = None
    S.sim_set_strategy_distance(C.c_float(first_call_distance))


def oracle_hot_path(planes, distance, force_dct8=False, keep=False, nthreads=1):
    """planes: [3, h, w] float32.  Returns HotPathResult (and the raw frame if keep).  nthreads > 1: the frame's
    256 x 256 groups over that many POSIX threads (orc_encode_hot_path_threads; same results)."""
    L = oracle()
    _, h, w = planes.shape
    f = OrcFrame()
    if nthreads > 1 and hasattr(L, "orc_encode_hot_path_threads"):
        rc = L.orc_encode_hot_path_threads(_plane_ptrs(planes), w, w, h, C.c_float(distance), int(force_dct8),
                                           int(nthreads), 1, C.byref(f))
    else:
        rc = L.orc_encode_hot_path(_plane_ptrs(planes), w, w, h, C.c_float(distance), int(force_dct8),
                                   C.byref(f))
    if rc != 0:
        raise ValueError("oracle rejected input: rc=%d" % rc)
    r = HotPathResult()
    r.xsize, r.ysize = w, h
    xb, yb, xt, yt = f.xsize_blocks, f.ysize_blocks, f.xsize_tiles, f.ysize_tiles
    ng = f.xsize_groups * f.ysize_groups
    r.quant_dc = np.stack([_np(f.quant_dc[c], (yb, xb), np.int16) for c in range(3)])
    r.raw_quant = _np(f.raw_quant_field, (yb, xb), np.uint8)
    r.strategy = _np(f.ac_strategy, (yb, xb), np.uint8)
    r.ytox = _np(f.ytox_map, (yt, xt), np.int8)
    r.ytob = _np(f.ytob_map, (yt, xt), np.int8)
    r.group_tokens = [C.string_at(f.group_tokens[g], f.group_token_bytes[g]) for g in range(ng)]
    r.xyb = np.stack([_np(f.xyb[c], (yb * 8, xb * 8), np.float32) for c in range(3)])
    r.qf = _np(f.quant_field, (yb, xb), np.float32)
    r.mask = _np(f.masking, (yb, xb), np.float32)
    r.ent8 = _np(f.entropy8, (yb // 2 + 1, xb // 2 + 1, 8), np.float32)
    if keep:
        return r, f
    L.orc_frame_free(C.byref(f))
    return r


# --------------------------------------------------------------------------- simulator
class SimResult(C.Structure):
    _fields_ = [("xsize_blocks", C.c_size_t), ("ysize_blocks", C.c_size_t),
                ("xsize_tiles", C.c_size_t), ("ysize_tiles", C.c_size_t), ("num_groups", C.c_size_t),
                ("quant_dc", C.POINTER(C.c_int16) * 3), ("raw_quant", C.POINTER(C.c_uint8)),
                ("strategy", C.POINTER(C.c_uint8)), ("ytox", C.POINTER(C.c_int8)),
                ("ytob", C.POINTER(C.c_int8)), ("tokens", C.POINTER(C.c_uint8)),
                ("group_tok_offset", C.POINTER(C.c_uint64)),
                ("xyb", fp * 3), ("qf", fp), ("mask", fp), ("ent8", fp),
                ("histogram", C.POINTER(C.c_uint32)), ("dc_records", C.POINTER(C.c_uint8)),
                ("dc_rec_offset", C.POINTER(C.c_uint64)), ("dc_count", C.POINTER(C.c_uint32)),
                ("num_dc_groups", C.c_size_t), ("exact_reruns", C.c_uint32), ("unsupported", C.c_uint32)]


_sim = None


_sim_tiny = None


def _sim_lib(tiny_root_table=False):
    global _sim, _sim_tiny
    if tiny_root_table:
        if _sim_tiny is None:
            _sim_tiny = C.CDLL(str(build_sim(True)))
            _sim_tiny.sim_encode.argtypes = [C.POINTER(fp), C.c_size_t, C.c_size_t, C.c_size_t, C.c_float,
                                             C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32,
                                             C.POINTER(SimResult)]
        return _sim_tiny
    if _sim is None:
        _sim = C.CDLL(str(build_sim()))
        _sim.sim_encode.argtypes = [C.POINTER(fp), C.c_size_t, C.c_size_t, C.c_size_t, C.c_float,
                                    C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32,
                                    C.POINTER(SimResult)]
        _sim.sim_pack_direct.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_int, C.c_void_p, C.c_void_p]
        _sim.sim_pack_deliver.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                          C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _sim.sim_pack_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p]
        _sim.sim_pack_lookback.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        _sim.sim_publish.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        _sim.sim_publish.restype = None
    return _sim


def hybrid_uint(value):
    """token.h:32-48 -> (symbol, nbits, extra bits)."""
    if value < 16:
        return value, 0, 0
    n = value.bit_length() - 1
    m = value - (1 << n)
    return (n << 2) + (m >> (n - 2)), n - 2, value & ((1 << (n - 2)) - 1)


def token_histogram(token_bytes):
    """[64][64] counts of (pre-clustered context, hybrid-uint symbol) of raw 3-byte records."""
    h = np.zeros((64, 64), np.uint32)
    a = np.frombuffer(token_bytes, np.uint8).reshape(-1, 3)
    for ctx, lo, hi in a:
        if ctx < 128:
            h[ctx, hybrid_uint(int(lo) | (int(hi) << 8))[0]] += 1
    return h


def pack_sections_python(sections, table):
    """Reference bit packer (enc_frame.cc:784-800): list of record bytes -> list of (bytes, nbits)."""
    out = []
    for rec in sections:
        acc, nbits = 0, 0
        a = np.frombuffer(rec, np.uint8).reshape(-1, 3)
        for ctx, lo, hi in a:
            value = int(lo) | (int(hi) << 8)
            if ctx >= 128:
                n, data = int(ctx) - 128, value
            else:
                sym, nb, extra = hybrid_uint(value)
                e = int(table[int(ctx) * 64 + sym])
                depth = e >> 16
                n, data = depth + nb, (e & 0xFFFF) | (extra << depth)
            acc |= data << nbits
            nbits += n
        out.append((acc.to_bytes((nbits + 7) // 8, "little"), nbits))
    return out


def sim_pack_sections(sections, table, misalign=0, nlaunch=1):
    """Tile-granular measure + write kernels on the CPU execution model (blob base = out + 4 *
    misalign words); also checks that no byte outside the sections' final places was written."""
    L = _sim_lib()
    blob = np.frombuffer(b"".join(sections) + b"\0\0\0\0", np.uint8).copy()
    offs = np.zeros(len(sections) + 1, np.uint64)
    offs[1:] = np.cumsum([len(x) // 3 for x in sections])
    table = np.ascontiguousarray(table, np.uint32)
    out = np.full(4 * int(offs[-1]) + 64, 0xCD, np.uint8)
    out_off = np.zeros(len(sections) + 1, np.uint64)
    out_bits = np.zeros(len(sections), np.uint32)
    L.sim_pack_direct(blob.ctypes.data, offs.ctypes.data, len(sections), table.ctypes.data, out.ctypes.data,
                      misalign, nlaunch, out_off.ctypes.data, out_bits.ctypes.data)
    total = int(out_off[-1])
    base = 4 * misalign
    # (the dword the last section ends in is zeroed before the writing pass: up to 3 bytes behind the blob)
    tail = (4 - (base + total) % 4) % 4
    assert (out[:base] == 0xCD).all() and (out[base + total + tail:] == 0xCD).all(), "stray stores"
    assert np.isin(out[base + total:base + total + tail], (0, 0xCD)).all(), "stray stores"
    body = out[base:base + total]
    return [(body[int(out_off[i]):int(out_off[i + 1])].tobytes(), int(out_bits[i])) for i in range(len(sections))]


def sim_pack_stream(sections, table, nlaunch=3):
    """The single pass on the CPU execution model: plan, then pack_tile_stream_kernel in `nlaunch` growing shares into a
    zeroed blob.  Returns ([(section bytes, section bits)], sections completed behind every launch); checks that
    nothing behind the last section (but the rest of its last dword) was written."""
    L = _sim_lib()
    rec = np.frombuffer(b"".join(sections) + b"\0\0\0\0", np.uint8).copy()
    offs = np.zeros(len(sections) + 1, np.uint64)
    offs[1:] = np.cumsum([len(x) // 3 for x in sections])
    table = np.ascontiguousarray(table, np.uint32)
    blob = np.zeros(4 * int(offs[-1]) + 8 * len(sections) + 256, np.uint8)
    bits = np.zeros(len(sections), np.uint32)
    launch_end = np.full(8, 0xABCD, np.uint32)
    assert L.sim_pack_stream(rec.ctypes.data, offs.ctypes.data, len(sections), table.ctypes.data, nlaunch, blob.ctypes.data,
                             bits.ctypes.data, launch_end.ctypes.data) == 0
    nbytes = (bits.astype(np.int64) + 7) // 8
    start = np.concatenate([[0], np.cumsum(nbytes)])
    assert not blob[(int(start[-1]) + 3) // 4 * 4:].any(), "stray stores behind the last section"
    out = [(blob[int(start[i]):int(start[i + 1])].tobytes(), int(bits[i])) for i in range(len(sections))]
    return out, [int(v) for v in launch_end[:nlaunch]]


def sim_pack_lookback(tile_states, block_states, tile, first):
    """The look-back of the single pass on the CPU execution model: the bit position at which `tile` starts, given
    the 64-bit states of the tiles in front of it in its block and of the blocks in front.  Returns (start, what the
    tile's block does to a position if the tile were its last and 5 bits long, as the 64-bit block state); every lane
    of the wave must have come up with the same."""
    L = _sim_lib()
    ts = np.ascontiguousarray(tile_states, np.uint64)
    bs = np.ascontiguousarray(block_states, np.uint64)
    out = np.zeros(128, np.uint64)
    assert L.sim_pack_lookback(ts.ctypes.data, bs.ctypes.data, tile, int(first), out.ctypes.data) == 0
    assert (out[:64] == out[0]).all() and (out[64:] == out[64]).all()
    return int(out[0]), int(out[64])


def sim_pack_deliver(sections, table, nlaunch=3, mode=0, shift=0, runs=None, grid=5):
    """Measure, write in growing shares and hand over (pack_deliver_kernel) on the CPU execution model.  mode 0: all
    sections back to back starting `shift` bytes into the destination; 1: ENDING there; 2: the runs (first section,
    count, destination offset) only.  Returns (destination bytes, where the hand-over starts, section byte offsets,
    section bits); the destination arrives poisoned with 0xCD."""
    L = _sim_lib()
    blob = np.frombuffer(b"".join(sections) + b"\0\0\0\0", np.uint8).copy()
    offs = np.zeros(len(sections) + 1, np.uint64)
    offs[1:] = np.cumsum([len(x) // 3 for x in sections])
    table = np.ascontiguousarray(table, np.uint32)
    cap = 4 * int(offs[-1]) + 8 * len(sections) + 4096 + shift
    if runs is not None:
        cap = max(cap, max(int(r[2]) for r in runs) + 4 * int(offs[-1]) + 4096)
    dst = np.full(cap, 0xCD, np.uint8)
    out_off = np.zeros(len(sections) + 1, np.uint64)
    out_bits = np.zeros(len(sections), np.uint32)
    flag = np.zeros(1, np.uint32)
    r = np.ascontiguousarray(np.array(runs if runs is not None else [[0, 0, 0]], np.uint64).reshape(-1))
    rc = L.sim_pack_deliver(blob.ctypes.data, offs.ctypes.data, len(sections), table.ctypes.data, nlaunch, mode,
                            dst.ctypes.data, shift, r.ctypes.data, 0 if runs is None else len(runs), grid,
                            out_off.ctypes.data, out_bits.ctypes.data, flag.ctypes.data)
    assert rc == 0 and int(flag[0]) == 77, "completion protocol of the hand-over"
    total = int(out_off[-1])
    start = shift - total if mode == 1 else shift
    return dst, start, out_off, out_bits


def pfm_payload(planes, big_endian=False):
    """The sample payload of a PFM file holding `planes` ([3, h, w] float32): interleaved RGB,
    bottom row first (read_pfm.cc:199-209), byte-swapped for the big-endian flavour."""
    a = np.ascontiguousarray(np.transpose(planes, (1, 2, 0))[::-1], np.float32)
    return a.astype(">f4" if big_endian else "<f4").view(np.float32).reshape(-1).copy()


def sim_hot_path(planes, distance, force_dct8=False, as_pfm=None, tiny_root_table=False, production_variant=False,
                 wide_token_index=False):
    """Runs the product's HIP kernels on the CPU execution model (tests only).  as_pfm = "le" /
    "be": the kernels read the frame from a raw PFM payload instead of planar planes.
    production_variant: tile_kernel as the product launches it (without the debug outputs xyb, qf,
    mask, ent8) instead of tile12_kernel_debug.  wide_token_index: token_kernel_wide (64-bit coefficient indices:
    what the product launches for frames above 1.43 Gpixel) instead of token_kernel."""
    _sim = _sim_lib(tiny_root_table)
    _, h, w = planes.shape
    p = distance_params(distance)
    s = SimResult()
    flags = (1 if force_dct8 else 0) | (0x800 if production_variant else 0) | (0x4000 if wide_token_index else 0)
    if as_pfm:
        payload = pfm_payload(planes, as_pfm == "be")
        ptrs = (fp * 3)(payload.ctypes.data_as(fp), None, None)
        flags |= 0x100 | (0x200 if as_pfm == "be" else 0)
    else:
        ptrs = _plane_ptrs(planes)
    rc = _sim.sim_encode(ptrs, w, w, h, p.distance, p.scale, p.inv_scale, p.scale_dc,
                         p.x_qm_scale, flags, C.byref(s))
    assert rc == 0
    r = HotPathResult()
    r.xsize, r.ysize = w, h
    xb, yb, xt, yt, ng = s.xsize_blocks, s.ysize_blocks, s.xsize_tiles, s.ysize_tiles, s.num_groups
    r.quant_dc = np.stack([_np(s.quant_dc[c], (yb, xb), np.int16) for c in range(3)])
    r.raw_quant = _np(s.raw_quant, (yb, xb), np.uint8)
    r.strategy = _np(s.strategy, (yb, xb), np.uint8)
    r.ytox = _np(s.ytox, (yt, xt), np.int8)
    r.ytob = _np(s.ytob, (yt, xt), np.int8)
    offs = [s.group_tok_offset[g] for g in range(ng + 1)]
    blob = C.string_at(s.tokens, offs[ng] * 3)
    r.group_tokens = [blob[3 * offs[g]:3 * offs[g + 1]] for g in range(ng)]
    r.exact_reruns = int(s.exact_reruns)
    r.unsupported = int(s.unsupported)
    r.xyb = np.stack([_np(s.xyb[c], (yb * 8, xb * 8), np.float32) for c in range(3)])
    r.qf = _np(s.qf, (yb, xb), np.float32)
    r.mask = _np(s.mask, (yb, xb), np.float32)
    r.ent8 = _np(s.ent8, (yb // 2 + 1, xb // 2 + 1, 8), np.float32)
    hh = _np(s.histogram, (2, 64, 64), np.uint32)
    r.histogram, r.dc_histogram = hh[0], hh[1]
    r.dc_records = [C.string_at(C.addressof(s.dc_records.contents) + 3 * s.dc_rec_offset[i], 3 * s.dc_count[i])
                    for i in range(s.num_dc_groups)]
    _sim.sim_free(C.byref(s))
    return r


def compare_results(a, b, what_a="A", what_b="B", check_debug=True):
    """Returns a list of human-readable mismatches (empty == bit-exact)."""
    bad = []

    def cmp(name, x, y, bits=False):
        if bits:
            x, y = x.view(np.uint32), y.view(np.uint32)
        if x.shape != y.shape:
            bad.append("%s: shape %s vs %s" % (name, x.shape, y.shape))
            return
        ne = np.argwhere(x != y)
        if len(ne):
            i = tuple(ne[0])
            bad.append("%s: %d mismatches, first at %s: %s=%r %s=%r" %
                       (name, len(ne), i, what_a, x[i], what_b, y[i]))

    if check_debug:
        cmp("xyb", a.xyb, b.xyb, bits=True)
        cmp("quant_field", a.qf, b.qf, bits=True)
        cmp("masking", a.mask, b.mask, bits=True)
    cmp("ytox", a.ytox, b.ytox)
    cmp("ytob", a.ytob, b.ytob)
    if check_debug:
        ea, eb = a.ent8.view(np.uint32), b.ent8.view(np.uint32)
        valid = ~np.isnan(a.ent8)
        if not np.array_equal(np.isnan(a.ent8), np.isnan(b.ent8)):
            bad.append("ent8: evaluated-cell sets differ")
        elif np.any(ea[valid] != eb[valid]):
            ne = np.argwhere((ea != eb) & valid)
            i = tuple(ne[0])
            bad.append("ent8: %d mismatches, first at %s: %r vs %r" % (len(ne), i, a.ent8[i], b.ent8[i]))
    cmp("ac_strategy", a.strategy, b.strategy)
    cmp("raw_quant", a.raw_quant, b.raw_quant)
    cmp("quant_dc", a.quant_dc, b.quant_dc)
    if len(a.group_tokens) != len(b.group_tokens):
        bad.append("group count differs")
    else:
        for g, (ta, tb) in enumerate(zip(a.group_tokens, b.group_tokens)):
            if ta != tb:
                n = min(len(ta), len(tb))
                first = next((i for i in range(n) if ta[i] != tb[i]), n)
                bad.append("tokens group %d: len %d vs %d, first diff at byte %d (token %d)" %
                           (g, len(ta), len(tb), first, first // 3))
                break
    return bad


# --------------------------------------------------------------------------- product
def product():
    """The libjxl-tiny_amd package (ctypes binding of the product libraries)."""
    sys.path.insert(0, str(ROOT))
    import __graft_entry__
    return __graft_entry__.load_package()


def host_dc_records(res):
    """Raw DC-group records of the host tokeniser (jxlt_debug_dc_records) for a HotPathResult."""
    P = product()
    H = P.host_lib()
    fr, keep = _frame_struct(res)
    ndc = ((res.xsize + 2047) // 2048) * ((res.ysize + 2047) // 2048)
    out = []
    for i in range(ndc):
        p, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        assert H.jxlt_debug_dc_records(C.byref(fr), i, C.byref(p), C.byref(n)) == 0
        out.append(C.string_at(p, n.value))
        H.jxlt_free(p)
    return out


_reference_single_symbol = False


@contextlib.contextmanager
def reference_single_symbol_codes():
    """While active both back-ends -- the oracle's (oracle_codestream / assemble_codestream) and the product's
    host library (jxl::EmulateReferenceSingleSymbolCodes) -- write one bit per token of a single-symbol prefix
    code, as the reference does: byte-identical to the reference even where its output cannot be decoded.
    Fixtures / known answers that pin REFERENCE bytes use this."""
    global _reference_single_symbol
    P = product()
    P.emulate_reference_single_symbol_codes(True)
    _reference_single_symbol = True
    try:
        yield
    finally:
        P.emulate_reference_single_symbol_codes(False)
        _reference_single_symbol = False


class OrcBsInput(C.Structure):
    _fields_ = [("xsize", C.c_size_t), ("ysize", C.c_size_t), ("quant_dc", C.POINTER(C.c_int16) * 3),
                ("raw_quant_field", C.POINTER(C.c_uint8)), ("ac_strategy", C.POINTER(C.c_uint8)),
                ("ytox_map", C.POINTER(C.c_int8)), ("ytob_map", C.POINTER(C.c_int8)),
                ("group_tokens", C.POINTER(C.POINTER(C.c_uint8))), ("group_token_bytes", C.POINTER(C.c_size_t))]


def _bs_input(res):
    """orc_bs_input over a HotPathResult (oracle, simulator or GPU); returns (struct, keepalive)."""
    b = OrcBsInput()
    b.xsize, b.ysize = res.xsize, res.ysize
    qd = [np.ascontiguousarray(res.quant_dc[c], np.int16) for c in range(3)]
    for c in range(3):
        b.quant_dc[c] = qd[c].ctypes.data_as(C.POINTER(C.c_int16))
    rq, st = np.ascontiguousarray(res.raw_quant, np.uint8), np.ascontiguousarray(res.strategy, np.uint8)
    tx, tb = np.ascontiguousarray(res.ytox, np.int8), np.ascontiguousarray(res.ytob, np.int8)
    b.raw_quant_field = rq.ctypes.data_as(C.POINTER(C.c_uint8))
    b.ac_strategy = st.ctypes.data_as(C.POINTER(C.c_uint8))
    b.ytox_map = tx.ctypes.data_as(C.POINTER(C.c_int8))
    b.ytob_map = tb.ctypes.data_as(C.POINTER(C.c_int8))
    n = len(res.group_tokens)
    bufs = [np.frombuffer(t + b"\0", np.uint8).copy() for t in res.group_tokens]
    ptrs = (C.POINTER(C.c_uint8) * max(n, 1))()
    lens = (C.c_size_t * max(n, 1))()
    for i, t in enumerate(res.group_tokens):
        ptrs[i] = bufs[i].ctypes.data_as(C.POINTER(C.c_uint8))
        lens[i] = len(t)
    b.group_tokens = C.cast(ptrs, C.POINTER(C.POINTER(C.c_uint8)))
    b.group_token_bytes = C.cast(lens, C.POINTER(C.c_size_t))
    return b, (qd, rq, st, tx, tb, bufs, ptrs, lens)


def _oracle_bs():
    L = oracle()
    L.orc_bs_encode_file.argtypes = [C.POINTER(OrcBsInput), C.c_float, C.c_int, C.POINTER(C.POINTER(C.c_uint8)),
                                     C.POINTER(C.c_size_t)]
    L.orc_bs_encode_file.restype = C.c_int
    L.orc_bs_dc_group_records.argtypes = [C.POINTER(OrcBsInput), C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)),
                                          C.POINTER(C.c_size_t)]
    L.orc_bs_dc_group_records.restype = C.c_int
    L.orc_bs_build_code_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_bs_build_code_tables.restype = None
    L.orc_bs_free.argtypes = [C.c_void_p]
    L.orc_bs_free.restype = None
    return L


def oracle_codestream(res, distance, reference_single_symbol=None):
    """Full .jxl bytes from a HotPathResult through the ORACLE's bitstream stage
    (oracle/jxl_tiny_bitstream_oracle.c): no product code involved."""
    L = _oracle_bs()
    b, keep = _bs_input(res)
    ref = _reference_single_symbol if reference_single_symbol is None else reference_single_symbol
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    rc = L.orc_bs_encode_file(C.byref(b), C.c_float(distance), int(ref), C.byref(out), C.byref(n))
    assert rc == 0, rc
    data = C.string_at(out, n.value)
    L.orc_bs_free(out)
    return data


def oracle_encode_file(planes, distance, nthreads=1, reference_single_symbol=None):
    """planes [3, h, w] float32 -> (.jxl bytes, seconds in the pixel pipeline, seconds in the bitstream stage), all
    inside the oracle's C code: orc_encode_hot_path_threads (the 256 x 256 groups over `nthreads` POSIX threads; 1 =
    the reference's own structure) and orc_bs_encode_file (serial, like the reference's OptimizeSections), with the
    frame handed from one to the other in place -- what bench.py times as cpu_baseline."""
    import time
    L, B = oracle(), _oracle_bs()
    _, h, w = planes.shape
    f = OrcFrame()
    t0 = time.perf_counter()
    rc = L.orc_encode_hot_path_threads(_plane_ptrs(planes), w, w, h, C.c_float(distance), 0, int(max(1, nthreads)),
                                       0, C.byref(f))
    if rc != 0:
        raise ValueError("oracle rejected input: rc=%d" % rc)
    t1 = time.perf_counter()
    b = OrcBsInput()
    b.xsize, b.ysize = w, h
    for c in range(3):
        b.quant_dc[c] = f.quant_dc[c]
    b.raw_quant_field, b.ac_strategy = f.raw_quant_field, f.ac_strategy
    b.ytox_map, b.ytob_map = f.ytox_map, f.ytob_map
    b.group_tokens, b.group_token_bytes = f.group_tokens, f.group_token_bytes
    ref = _reference_single_symbol if reference_single_symbol is None else reference_single_symbol
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    rc = B.orc_bs_encode_file(C.byref(b), C.c_float(distance), int(ref), C.byref(out), C.byref(n))
    t2 = time.perf_counter()
    assert rc == 0, rc
    data = C.string_at(out, n.value)
    B.orc_bs_free(out)
    L.orc_frame_free(C.byref(f))
    return data, t1 - t0, t2 - t1


def oracle_dc_records(res):
    """Raw DC-group records (WriteDCGroup, OPTIMIZE_CODE form) from the oracle's bitstream stage."""
    L = _oracle_bs()
    b, keep = _bs_input(res)
    ndc = ((res.xsize + 2047) // 2048) * ((res.ysize + 2047) // 2048)
    out = []
    for i in range(ndc):
        p, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        assert L.orc_bs_dc_group_records(C.byref(b), i, C.byref(p), C.byref(n)) == 0
        out.append(C.string_at(p, n.value))
        L.orc_bs_free(p)
    return out


def oracle_code_tables(ac_hist, dc_hist, reference_single_symbol=None):
    """(ac_table, dc_table) uint32[4096] each from [64][64] histograms, by the oracle's clustering + Huffman."""
    L = _oracle_bs()
    ref = _reference_single_symbol if reference_single_symbol is None else reference_single_symbol
    a = np.ascontiguousarray(ac_hist, np.uint32).reshape(-1)
    d = np.ascontiguousarray(dc_hist, np.uint32).reshape(-1)
    at, dt = np.zeros(4096, np.uint32), np.zeros(4096, np.uint32)
    L.orc_bs_build_code_tables(a.ctypes.data, d.ctypes.data, int(ref), at.ctypes.data, dt.ctypes.data)
    return at, dt


def assemble_codestream(res, distance, num_threads=1):
    """Full .jxl bytes for a HotPathResult: the checker's codestream, produced by the oracle's own
    bitstream stage (independent of libjxltiny_host.so)."""
    return oracle_codestream(res, distance)


def host_assemble_codestream(res, distance, num_threads=1):
    """Full .jxl bytes from a HotPathResult via the PRODUCT's host back-end
    (jxlt_write_file_header + jxlt_assemble_frame) -- the thing under test, not a checker."""
    P = product()
    H = P.host_lib()
    dp = P.distance_params(distance)
    fr, keep = _frame_struct(res)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    rc = H.jxlt_assemble_frame(C.byref(fr), C.byref(dp), num_threads, C.byref(out), C.byref(n))
    assert rc == 0, rc
    frame = C.string_at(out, n.value)
    H.jxlt_free(out)
    return P.file_header(res.xsize, res.ysize) + frame


def _frame_struct(res):
    P = product()
    fr = P.FrameResult()
    fr.xsize, fr.ysize = res.xsize, res.ysize
    yb, xb = res.raw_quant.shape
    yt, xt = res.ytox.shape
    fr.xsize_blocks, fr.ysize_blocks, fr.xsize_tiles, fr.ysize_tiles = xb, yb, xt, yt
    fr.num_groups = len(res.group_tokens)
    keep = [np.ascontiguousarray(res.quant_dc[c]) for c in range(3)]
    for c in range(3):
        fr.quant_dc[c] = keep[c].ctypes.data_as(C.POINTER(C.c_int16))
    rq, st = np.ascontiguousarray(res.raw_quant), np.ascontiguousarray(res.strategy)
    tx, tb = np.ascontiguousarray(res.ytox), np.ascontiguousarray(res.ytob)
    fr.raw_quant_field = rq.ctypes.data_as(C.POINTER(C.c_uint8))
    fr.ac_strategy = st.ctypes.data_as(C.POINTER(C.c_uint8))
    fr.ytox_map = tx.ctypes.data_as(C.POINTER(C.c_int8))
    fr.ytob_map = tb.ctypes.data_as(C.POINTER(C.c_int8))
    blob = np.frombuffer(b"".join(res.group_tokens) + b"\0", dtype=np.uint8).copy()
    offs = np.zeros(fr.num_groups + 1, np.uint64)
    offs[1:] = np.cumsum([len(t) for t in res.group_tokens])
    fr.tokens = blob.ctypes.data_as(C.POINTER(C.c_uint8))
    fr.group_token_offset = offs.ctypes.data_as(C.POINTER(C.c_uint64))
    return fr, (keep, rq, st, tx, tb, blob, offs)
