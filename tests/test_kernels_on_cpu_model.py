"""Runs the product's HIP kernels (libjxl-tiny_amd/csrc/jxlt_device.h) on the
fiber-based CPU execution model in tests/hipsim and requires bit-exact equality
with the oracle for every intermediate and output.  This is test infrastructure
only -- the product never runs on the CPU -- and lets kernel changes be checked
in a container without a GPU; the `-m gpu` tests repeat the comparison on a real
MI355X through the C ABI."""
import numpy as np
import pytest

import jxlt_testlib as T

CASES = [
    # w, h, distance, hard, force_dct8
    (96, 72, 1.0, False, False),    # partial tiles in both directions
    (9, 7, 1.0, False, False),      # two blocks, odd sizes
    (200, 137, 1.0, False, False),  # odd size, several tiles
    (300, 264, 2.0, False, False),  # 2x2 groups, x_qm_scale 3
    (64, 64, 0.5, False, False),
    (130, 70, 8.0, False, False),   # dampened modulation, epf
    (72, 72, 16.0, False, False),   # x_qm_scale 4
    (128, 64, 1.0, True, False),    # uniform noise: token heavy
    (128, 64, 1.0, False, True),    # fixed DCT8 (BASELINE config #2 mode)
    (17, 130, 3.0, False, False),   # narrow
    (24, 2048 + 2048 + 72, 1.0, False, False),  # three rows of DC groups: the row-wise launches (slab arguments,
                                                # chained token-offset scan, per-row DC kernels) of jxlt_capi_encode.hip
]


@pytest.mark.parametrize("w,h,distance,hard,dct8", CASES)
def test_kernels_match_oracle_bit_exact(built, w, h, distance, hard, dct8):
    planes = T.to_planes(T.synthetic_image(w, h, hard=hard))
    want = T.oracle_hot_path(planes, distance, dct8)
    got = T.sim_hot_path(planes, distance, dct8)
    assert T.compare_results(want, got, "oracle", "kernels") == []


@pytest.mark.parametrize("w,h,distance,hard,dct8", [CASES[0], CASES[1], CASES[3], CASES[5], CASES[7]])
def test_production_variant_of_tile_kernel_matches_oracle(built, w, h, distance, hard, dct8):
    """tile_kernel as the product launches it (compiled without the debug outputs the test above
    reads through tile_kernel_debug): everything that reaches the codestream, bit-exact."""
    planes = T.to_planes(T.synthetic_image(w, h, hard=hard))
    want = T.oracle_hot_path(planes, distance, dct8)
    got = T.sim_hot_path(planes, distance, dct8, production_variant=True)
    assert T.compare_results(want, got, "oracle", "kernels", check_debug=False) == []


@pytest.mark.parametrize("w,h,distance,hard,dct8", [CASES[0], CASES[3], CASES[7]])
def test_wide_index_variant_of_token_kernel_matches_oracle(built, w, h, distance, hard, dct8):
    """token_kernel_wide forms coefficient indices in 64 bits; the product launches it for frames above 2^32 / 192
    blocks (1.43 Gpixel: `-m gpu` has one), here it runs on ordinary frames."""
    planes = T.to_planes(T.synthetic_image(w, h, hard=hard))
    want = T.oracle_hot_path(planes, distance, dct8)
    got = T.sim_hot_path(planes, distance, dct8, wide_token_index=True)
    assert T.compare_results(want, got, "oracle", "kernels") == []


def test_token_kernel_histogram_matches_tokens(built):
    planes = T.to_planes(T.synthetic_image(300, 264))
    got = T.sim_hot_path(planes, 1.0)
    want = T.token_histogram(got.all_tokens())
    assert (got.histogram == want).all()
    assert int(got.histogram.sum()) == len(got.all_tokens()) // 3


def _random_code_table(rng):
    """A valid-looking table: arbitrary depths 1..15 and bit patterns below 2^depth."""
    depth = rng.integers(1, 16, size=64 * 64)
    bits = rng.integers(0, 1 << 15, size=64 * 64) & ((1 << depth) - 1)
    return ((depth << 16) | bits).astype("uint32")


@pytest.mark.parametrize("sizes", [[0], [1], [5, 0, 17], [4095, 4096, 4097], [9000, 3], [300] * 7, [0, 13000, 0, 2, 8200]])
def test_pack_kernel_matches_reference_packer(built, sizes):
    import numpy as np
    rng = np.random.default_rng(sum(sizes) + len(sizes))
    table = _random_code_table(rng)
    sections = []
    for n in sizes:
        ctx = rng.integers(0, 64, size=n).astype(np.uint8)
        # mix of small values, large values and raw-bit escapes (ctx >= 128)
        val = np.where(rng.random(n) < 0.8, rng.integers(0, 20, size=n), rng.integers(0, 65536, size=n))
        esc = rng.random(n) < 0.02
        nb = rng.integers(1, 17, size=n)
        ctx = np.where(esc, 128 + nb, ctx).astype(np.uint8)
        val = np.where(esc, val & ((1 << nb) - 1), val).astype(np.uint16)
        rec = np.stack([ctx, (val & 0xFF).astype(np.uint8), (val >> 8).astype(np.uint8)], axis=1)
        sections.append(rec.astype(np.uint8).tobytes())
    want = T.pack_sections_python(sections, table)
    # measure, then store at the final bit positions; two blob bases, one and two launches
    for words in range(2):
        assert T.sim_pack_sections(sections, table, words, nlaunch=1 + words) == want


def _random_sections(rng, sizes):
    import numpy as np
    sections = []
    for n in sizes:
        ctx = rng.integers(0, 64, size=n).astype(np.uint8)
        val = np.where(rng.random(n) < 0.8, rng.integers(0, 20, size=n), rng.integers(0, 65536, size=n)).astype(np.uint16)
        rec = np.stack([ctx, (val & 0xFF).astype(np.uint8), (val >> 8).astype(np.uint8)], axis=1)
        sections.append(rec.astype(np.uint8).tobytes())
    return sections


@pytest.mark.parametrize("sizes", [[0], [7], [5, 0, 17], [4095, 4096, 4097, 1], [0, 0, 0], [20000, 3, 9000, 0, 12345],
                                   [600] * 23])
@pytest.mark.parametrize("nlaunch", [1, 3, 4])
def test_hand_over_by_launches_delivers_every_section_once(built, sizes, nlaunch):
    """The hand-over behind the writing launches -- behind every launch the copy of the sections
    pack_tile_finalize_kernel has filed as complete (launch_sec_end; the product issues copy commands, the model a
    memcpy): whatever the shares of the launches, every section arrives exactly once, byte for byte what the reference
    packer gives, at any alignment of the destination, start- and end-aligned; nothing else is written."""
    import numpy as np
    rng = np.random.default_rng(len(sizes) * 1000 + sum(sizes) + nlaunch)
    table = _random_code_table(rng)
    sections = _random_sections(rng, sizes)
    want = T.pack_sections_python(sections, table)
    want_bytes = b"".join(b for b, _ in want)
    for mode, shift, grid in [(0, 0, 5), (0, 3, 1), (0, 21, 64), (1, 100000 + 7, 3), (1, 200001, 16)]:
        if mode == 1 and shift < len(want_bytes):
            continue
        dst, start, off, bits = T.sim_pack_deliver(sections, table, nlaunch=nlaunch, mode=mode, shift=shift, grid=grid)
        assert [int(b) for b in bits] == [nb for _, nb in want]
        assert int(off[-1]) == len(want_bytes)
        assert dst[start:start + len(want_bytes)].tobytes() == want_bytes
        assert (dst[:start] == 0xCD).all() and (dst[start + len(want_bytes):] == 0xCD).all(), "stray stores"


@pytest.mark.parametrize("seed", range(12))
def test_single_pass_look_back_composes_sizes_and_section_starts(built, seed):
    """The look-back of the single pass: where a tile starts follows from the SIZES of the tiles in front of it in
    its block of 64, the nearest block in front that knows its END and what the blocks between do to a position -- a
    size moves the position on, a tile that starts a section rounds it up to a byte first.  (On the CPU model
    workgroups run one after the other, so inside sim_pack_stream every tile finds the END of the block right in
    front of it; here the states are made up: ends at any distance -- in the first window of 64 blocks, windows
    back, none at all (block 0's position) --, section starts anywhere, states behind the nearest END holding junk.)"""
    rng = np.random.default_rng(7000 + seed)
    SIZE, END, FIRST = 1 << 62, 2 << 62, 1 << 61
    ntiles = int(rng.choice([1, 2, 5, 63, 64, 65, 255, 256, 257, 300, 4096 + 70, 64 * 150 + 3]))
    bits = rng.integers(0, 1 << 17, ntiles)
    bits[rng.random(ntiles) < 0.1] = 0
    first = rng.random(ntiles) < [0.0, 0.03, 0.3, 1.0][seed % 4]
    first[0] = True
    # the truth: positions tile by tile
    start = np.zeros(ntiles + 1, np.uint64)
    pos = 0
    for t in range(ntiles):
        if first[t]:
            pos = (pos + 7) & ~7
        start[t] = pos
        pos += int(bits[t])
    start[ntiles] = pos

    def run_of(t0, t1):  # what tiles t0 .. t1 - 1 do to a position: (rounds, pre, rest)
        rounds, pre, rest = 0, 0, 0
        for t in range(t0, t1):
            if first[t]:
                rest = ((rest + 7) & ~7) + int(bits[t]) if rounds else int(bits[t])
                rounds = 1
            elif rounds:
                rest += int(bits[t])
            else:
                pre += int(bits[t])
        return rounds, pre, rest

    nblocks = (ntiles + 63) // 64
    for trial in range(6):
        tile = ntiles - 1 if trial == 0 else int(rng.integers(0, ntiles))
        block = tile // 64
        # the nearest block that knows its end: none (trial 1), or `gap` blocks in front
        gap = block + 1 if trial == 1 else int(rng.integers(1, block + 2))
        tile_states = np.zeros(ntiles + 1, np.uint64)
        for t in range(ntiles):
            # (only the tiles in front of `tile` in its own block are looked at: everything else is junk)
            if block * 64 <= t < tile:
                tile_states[t] = SIZE | (FIRST if first[t] else 0) | int(bits[t])
            else:
                tile_states[t] = int(rng.choice([0, SIZE | 12345, SIZE | FIRST | 7]))
        block_states = np.zeros(nblocks + 1, np.uint64)
        for b in range(nblocks):
            if b == block - gap:
                block_states[b] = END | int(start[64 * b + 64])
            elif block - gap < b < block:
                rounds, pre, rest = run_of(64 * b, 64 * b + 64)
                block_states[b] = SIZE | (rounds << 61) | (pre << 31) | rest
            else:
                block_states[b] = int(rng.choice([0, SIZE | 12345, END | 99, SIZE | FIRST | 7]))
        got, whole = T.sim_pack_lookback(tile_states, block_states, tile, bool(first[tile]))
        assert got == int(start[tile]), (ntiles, tile, gap)
        # ... and what the block would tell the blocks behind it if this tile closed it with 5 bits of its own
        rounds, pre, rest = run_of(64 * block, tile)
        if first[tile]:
            rest = ((rest + 7) & ~7) + 5 if rounds else 5
            rounds = 1
        elif rounds:
            rest += 5
        else:
            pre += 5
        assert whole == SIZE | (rounds << 61) | (pre << 31) | rest


@pytest.mark.parametrize("sizes", [[0], [7], [5, 0, 17], [4095, 4096, 4097, 1], [0, 0, 0], [20000, 3, 9000, 0, 12345],
                                   [600] * 23, [1] * 300, [0, 4096 * 3, 0, 0, 1]])
@pytest.mark.parametrize("nlaunch", [1, 3, 4])
def test_single_pass_packing_matches_reference_packer(built, sizes, nlaunch):
    """pack_tile_stream_kernel (round 4): no measuring pass -- a tile takes its bit position from the tiles in front
    of it (sizes of the ones that do not know their end yet, byte rounding where a section begins) and the sections'
    bit counts are summed by their tiles.  Sections of every size incl. empty ones and one-record ones (many section
    starts inside one look-back window), short codes (tiles of a few bits), every launch split."""
    import numpy as np
    rng = np.random.default_rng(len(sizes) * 77 + sum(sizes) + nlaunch)
    for table in (_random_code_table(rng), ((np.ones(4096, np.int64) << 16) | rng.integers(0, 2, size=4096)).astype("uint32")):
        sections = _random_sections(rng, sizes)
        want = T.pack_sections_python(sections, table)
        got, launch_end = T.sim_pack_stream(sections, table, nlaunch=nlaunch)
        assert got == want
        # the sections every launch completes: non-decreasing, and the last launch that has tiles completes them all
        ends = [e for e in launch_end if e != 0xFFFFFFFF]
        assert ends == sorted(ends) and all(e <= len(sizes) for e in ends)
        if sum(sizes):
            last_nonempty = max(i for i, n in enumerate(sizes) if n)
            assert ends and ends[-1] == last_nonempty + 1


def test_hand_over_in_runs(built):
    """Run mode (a slab of a frame that several GPUs share owns several ranges of the codestream): runs of
    consecutive sections go to the offsets the caller names."""
    import numpy as np
    rng = np.random.default_rng(99)
    table = _random_code_table(rng)
    sizes = [int(v) for v in rng.integers(0, 900, size=130)]
    sections = _random_sections(rng, sizes)
    want = T.pack_sections_python(sections, table)
    lens = [len(b) for b, _ in want]
    # runs of 1..3 sections, scattered: run k goes to k * 4000 + k (odd alignments)
    runs, s = [], 0
    while s < len(sections):
        n = min(len(sections) - s, 1 + len(runs) % 3)
        runs.append([s, n, len(runs) * 4001 + 13])
        s += n
    assert len(runs) > 48
    dst, start, off, bits = T.sim_pack_deliver(sections, table, nlaunch=2, mode=2, shift=0, runs=runs, grid=4)
    touched = np.zeros(len(dst), bool)
    for first, n, at in runs:
        chunk = b"".join(want[i][0] for i in range(first, first + n))
        assert dst[at:at + len(chunk)].tobytes() == chunk
        touched[at:at + len(chunk)] = True
    assert (dst[~touched] == 0xCD).all(), "stray stores"


def test_publish_kernel(built):
    import ctypes as C
    import numpy as np
    L = T._sim_lib()
    a = np.arange(5000, dtype=np.uint32)
    b = np.arange(3, dtype=np.uint32) + 9
    da, db = np.zeros(5000, np.uint32), np.zeros(3, np.uint32)
    s64, d64, flag = np.array([2 ** 40 + 5], np.uint64), np.zeros(1, np.uint64), np.zeros(1, np.uint32)
    L.sim_publish(a.ctypes.data, da.ctypes.data, 5000, b.ctypes.data, db.ctypes.data, 3, s64.ctypes.data, d64.ctypes.data,
                  flag.ctypes.data, 41)
    assert (da == a).all() and (db == b).all() and int(d64[0]) == 2 ** 40 + 5 and int(flag[0]) == 41


def test_pack_tiles_with_short_codes(built):
    """Tile boundaries inside a dword: the later tile re-derives its predecessors' trailing bits.
    One-bit codes and zero-length escapes make many records share one dword, also across more
    than one 64-record look-back round and across a whole tile."""
    import numpy as np
    rng = np.random.default_rng(7)
    depth = np.ones(64 * 64, np.int64)
    bits = rng.integers(0, 2, size=64 * 64)
    table = ((depth << 16) | bits).astype("uint32")
    sections = []
    for n, p_zero in [(3 * 4096 + 77, 0.0), (2 * 4096 + 5, 0.97), (4096 + 1, 1.0), (4096, 0.5)]:
        ctx = rng.integers(0, 64, size=n).astype(np.uint8)
        val = rng.integers(0, 16, size=n).astype(np.uint16)  # no extra bits: 1 bit per coded record
        zero = rng.random(n) < p_zero                          # escape with 0 raw bits
        ctx = np.where(zero, 128, ctx).astype(np.uint8)
        val = np.where(zero, 0, val).astype(np.uint16)
        rec = np.stack([ctx, (val & 0xFF).astype(np.uint8), (val >> 8).astype(np.uint8)], axis=1)
        sections.append(rec.astype(np.uint8).tobytes())
    want = T.pack_sections_python(sections, table)
    assert T.sim_pack_sections(sections, table, 0, nlaunch=3) == want


@pytest.mark.parametrize("w,h,distance", [(200, 137, 1.0), (9, 7, 1.0), (300, 264, 2.0), (2100, 40, 1.0), (64, 64, 8.0),
                                          (2060, 2100, 2.0)])
def test_dc_kernels_match_oracle_tokeniser(built, w, h, distance):
    """dc_elementwise_kernel + dc_chain_kernel vs the oracle's WriteDCGroup restatement."""
    planes = T.to_planes(T.synthetic_image(w, h))
    # (the 2 x 2 DC groups of the large frame are a thousand tiles on the fiber model: there the tile kernel that
    # feeds the DC kernels is the production build, without the debug outputs)
    big = w * h > 1 << 20
    got = T.sim_hot_path(planes, distance, production_variant=big)
    want = T.oracle_dc_records(got)
    assert got.dc_records == want
    h = sum((T.token_histogram(r) for r in want))
    assert (got.dc_histogram == h).all()


@pytest.mark.parametrize("flavour", ["le", "be"])
def test_pfm_payload_ingest_on_cpu_model(built, flavour):
    """tile_kernel reading the frame straight from a raw PFM payload (interleaved, bottom-up,
    either byte order) produces what it produces from the planar frame."""
    planes = T.to_planes(T.synthetic_image(200, 137))
    a = T.sim_hot_path(planes, 1.0)
    b = T.sim_hot_path(planes, 1.0, as_pfm=flavour)
    T.compare_results(a, b, "planar", "pfm payload")


def test_root_table_overflow_redoes_the_tiles_concerned_with_computed_roots(built):
    """tile_kernel takes the square roots of the entropy estimate from a table; a tile with a quantised
    magnitude beyond it must file itself and be redone by tile*_kernel_redo with computed roots -- only that tile.
    Magnitudes >= 1024 hardly occur (the adaptive quantiser sees to that), so this runs a build of the kernels
    whose table has 16 entries: the busy tiles of an ordinary image overflow it, the flat ones do not."""
    img = T.synthetic_image(200, 137, hard=True)
    img[:64, :64] = 0.25  # a flat tile: every quantised coefficient is 0
    planes = T.to_planes(img)
    got = T.sim_hot_path(planes, 0.5, tiny_root_table=True)
    ntiles = 4 * 3
    assert 0 < got.exact_reruns < ntiles, "tiles redone: %d" % got.exact_reruns
    want = T.oracle_hot_path(planes, 0.5)
    assert T.compare_results(want, got, "oracle", "cpu model, computed roots") == []
    # the product build stays on the table path for the same image, with the same result
    same = T.sim_hot_path(planes, 0.5)
    assert same.exact_reruns == 0
    assert T.compare_results(want, same, "oracle", "cpu model, table roots") == []


def test_static_constant_emulation_on_cpu_model(built):
    """The reference freezes the transform search's two multipliers at the distance of the first
    frame of the process (enc_ac_strategy.cc:178-185).  With that emulated (first distance 16,
    this frame 0.5) oracle and kernels still agree, and the result differs from the unemulated one
    (otherwise the test would not test anything)."""
    planes = T.to_planes(T.synthetic_image(200, 137))
    plain = T.oracle_hot_path(planes, 0.5)
    T.set_strategy_distance(16.0)
    try:
        want = T.oracle_hot_path(planes, 0.5)
        got = T.sim_hot_path(planes, 0.5)
    finally:
        T.set_strategy_distance(0.0)
    assert T.compare_results(want, got, "oracle", "cpu model") == []
    assert (want.strategy != plain.strategy).any()


@pytest.mark.parametrize("n", [1, 7, 1023, 1024, 1025, 4096, 5000, 16448, 32768, 32769, 65536, 100003])
def test_group_scan_kernel_matches_cumsum(built, n):
    """Section bookkeeping of the packing stage: exclusive scan of 32-bit counts into 64-bit offsets, for every
    run length per thread the kernel can meet (1024 threads, up to 32 counts each per pass; more than 32 768
    counts -- narrow and tall frames: 64 x 16M pixels have 65 536 AC groups -- take several passes), with totals
    beyond 2^32."""
    import ctypes as C

    import numpy as np
    L = T._sim_lib()
    rng = np.random.default_rng(n)
    counts = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    got = np.full(n + 1, 0xDEAD, np.uint64)
    L.sim_group_scan.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.sim_group_scan.restype = None
    L.sim_group_scan(counts.ctypes.data, n, got.ctypes.data)
    want = np.concatenate([[0], np.cumsum(counts.astype(np.uint64))]).astype(np.uint64)
    assert (got == want).all()


def _poisoned_frame(poison, seed=7):
    """136 x 72: forty samples of the smooth test frame replaced by `poison`."""
    img = T.synthetic_image(136, 72, seed=3)
    idx = np.random.default_rng(seed).integers(0, img.size, size=40)
    img.reshape(-1)[idx] = np.float32(poison)
    return T.to_planes(img)


@pytest.mark.parametrize("poison", [1e38, -1e38, float("inf"), float("nan")], ids=["1e38", "-1e38", "inf", "nan"])
def test_values_the_format_cannot_carry_are_refused_or_leave_a_decodable_stream(built, poison):
    """A quantised coefficient whose token does not fit the format's 16 bits (the reference only asserts it in debug
    builds, enc_bit_writer.cc:120, and otherwise writes a stream no decoder accepts) or a DC value beyond int16 must
    be COUNTED by the kernels (TileArgs::unsupported: the C ABI answers JXLT_ERR_UNSUPPORTED) -- or the tokens must
    still be a well-formed stream.  Samples of +1e38 and +Inf overflow and must be counted (-1e38 is clamped to zero by
    the colour transform, enc_xyb.cc:73-75, and is an ordinary frame)."""
    import jxl_decoder as D
    got = T.sim_hot_path(_poisoned_frame(poison), 1.0)
    if poison > 1e30:
        assert got.unsupported > 0
    if got.unsupported == 0:
        D.decode(T.oracle_codestream(got, 1.0))  # (raises on a malformed stream)


def test_ordinary_frames_are_not_refused(built):
    """... and nothing is counted for frames the format can carry: HDR noise up to 40 at distance 0.05 (quantised
    magnitudes of several hundred), also when the tiles go through tile*_kernel_redo, where every value is tested
    exactly (the build with a 16-entry root table: every busy tile is redone)."""
    img = (T.synthetic_image(136, 72, seed=5, hard=True) * np.float32(40.0)).astype(np.float32)
    planes = T.to_planes(img)
    want = T.oracle_hot_path(planes, 0.05)
    got = T.sim_hot_path(planes, 0.05)
    assert got.unsupported == 0
    assert T.compare_results(want, got, "oracle", "cpu model") == []
    redone = T.sim_hot_path(planes, 0.05, tiny_root_table=True)
    assert redone.unsupported == 0 and redone.exact_reruns > 0
    assert T.compare_results(want, redone, "oracle", "cpu model, computed roots") == []
    # (the same noise at distance 0.02 is beyond the format: its quantised DC values leave int16)
    assert T.sim_hot_path(planes, 0.02).unsupported > 0
