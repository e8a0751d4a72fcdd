"""Parity tests proper: the HIP path on a real MI355X, called through the C ABI
(libjxltiny_hip.so / libjxltiny_host.so), against the CPU oracle on the same
seeded inputs.  Bar: bit-exact -- XYB / quant-field / masking / entropy floats to
0 ULP (the kernels implement the oracle's arithmetic model operation by
operation), every integer output and the token stream byte for byte."""
import subprocess

import ctypes as C

import numpy as np
import pytest

import jxlt_testlib as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def enc(built):
    e = built.Encoder(0)
    yield e
    e.close()


CASES = [
    # w, h, distance, hard, force_dct8
    (256, 256, 1.0, False, False),    # BASELINE config #1: one group
    (9, 7, 1.0, False, False),
    (96, 72, 1.0, False, False),
    (200, 137, 1.0, False, False),
    (300, 264, 2.0, False, False),
    (64, 64, 0.5, False, False),
    (130, 70, 8.0, False, False),
    (72, 72, 16.0, False, False),
    (65, 65, 0.1, True, False),
    (17, 300, 3.0, False, False),
    (512, 512, 1.0, True, False),     # uniform noise, token heavy
    (1024, 1024, 1.0, False, True),   # fixed DCT8 (config #2 mode)
    (1024, 1024, 1.0, False, False),
    (2100, 300, 1.0, False, False),   # two DC groups wide
    (520, 2100, 4.0, False, False),   # two DC groups tall, bottom group 52 px
    (1920, 1080, 1.0, False, False),  # 135-group frame shape of config #5 at half size
]


@pytest.mark.parametrize("w,h,distance,hard,dct8", CASES)
def test_hot_path_bit_exact_vs_oracle(enc, w, h, distance, hard, dct8):
    planes = T.to_planes(T.synthetic_image(w, h, hard=hard))
    want = T.oracle_hot_path(planes, distance, dct8)
    got = enc.hot_path(planes, distance, force_dct8=dct8, debug=True)
    assert T.compare_results(want, got, "oracle", "gpu") == []


@pytest.mark.parametrize("w,h,distance", [(256, 256, 1.0), (200, 137, 0.5), (700, 520, 2.0), (2100, 300, 1.0)])
def test_dropin_encode_file_matches_oracle_codestream(built, w, h, distance):
    planes = T.to_planes(T.synthetic_image(w, h))
    want = T.assemble_codestream(T.oracle_hot_path(planes, distance), distance)
    assert built.encode_file(planes, distance) == want


@pytest.mark.parametrize("big_endian", [False, True])
@pytest.mark.parametrize("host_ingest", [False, True])
def test_cjxl_tiny_cli(built, tmp_path, host_ingest, big_endian):
    """The command line tool, with the PFM payload ingested by the device kernels (default) or
    de-interleaved on the host like the reference's ReadPFM (--host-ingest); both byte orders."""
    img = T.synthetic_image(300, 200)
    pfm, out = tmp_path / "in.pfm", tmp_path / "out.jxl"
    T.write_pfm(pfm, img, big_endian)
    cmd = [str(built.CJXL_TINY), str(pfm), str(out), "-d", "1.5"] + (["--host-ingest"] if host_ingest else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Read 300x200 pixels input image." in r.stderr
    want = T.assemble_codestream(T.oracle_hot_path(T.to_planes(img), 1.5), 1.5)
    assert out.read_bytes() == want
    assert ("Compressed to %d bytes." % len(want)) in r.stderr


@pytest.mark.parametrize("big_endian", [False, True])
def test_the_references_own_cjxl_main_runs_on_the_product(built, tmp_path, big_endian):
    """The reference's cjxl_main.cc, unmodified, compiled against the drop-in headers and linked with the product's
    host library (oracle/Makefile: ref-caller; built in the container that has /root/reference, the binary travels):
    PFM in, oracle's bytes out, the reference's own messages."""
    exe = T.ROOT / "oracle" / "_ref" / "cjxl_tiny_ref_main"
    if not exe.exists():
        pytest.skip("oracle/_ref/cjxl_tiny_ref_main was not built (no reference checkout where build() ran)")
    img = T.synthetic_image(333, 270, seed=77)
    pfm, out = tmp_path / "in.pfm", tmp_path / "out.jxl"
    T.write_pfm(pfm, img, big_endian)
    r = subprocess.run([str(exe), str(pfm), str(out), "-d", "0.8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = T.assemble_codestream(T.oracle_hot_path(T.to_planes(img), 0.8), 0.8)
    assert out.read_bytes() == want
    assert "Read 333x270 pixels input image." in r.stderr and ("Compressed to %d bytes." % len(want)) in r.stderr


@pytest.mark.parametrize("host_ingest", [False, True])
def test_cjxl_tiny_over_a_device_list(built, tmp_path, host_ingest):
    """JXLT_DEVICES: the unmodified command line spreads one frame over several device contexts (here GPU 0
    twice): PFM payload slabs straight to the devices, or ReadPFM + EncodeFile (--host-ingest)."""
    import os
    img = T.synthetic_image(300, 2048 + 260)
    pfm, out = tmp_path / "in.pfm", tmp_path / "out.jxl"
    T.write_pfm(pfm, img, big_endian=True)
    cmd = [str(built.CJXL_TINY), str(pfm), str(out), "-d", "2"] + (["--host-ingest"] if host_ingest else [])
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, JXLT_DEVICES="0,0"))
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == T.assemble_codestream(T.oracle_hot_path(T.to_planes(img), 2.0), 2.0)


def test_error_behaviour(built, enc):
    # EncodeFile rejects distance <= 0 (enc_file.cc:57-62) and empty images
    planes = T.to_planes(T.synthetic_image(32, 32))
    for bad in (0.0, -1.0):
        with pytest.raises(built.JxlTinyError):
            built.encode_file(planes, bad)
    # distance <= 0.03 is clamped to 0.03 (enc_file.cc:63-65)
    assert built.encode_file(planes, 0.01) == built.encode_file(planes, 0.03)
    # images that fit one 8x8 block trap in the reference; here: an error
    with pytest.raises(built.JxlTinyError):
        enc.upload(np.zeros((3, 8, 8), np.float32))
    # calls out of order say so instead of reading what is not there
    hip = built.hip_lib()
    fresh = built.Encoder(0)
    table = np.zeros(4096, np.uint32)
    hip.jxlt_histograms_ready.argtypes = [C.c_void_p]
    hip.jxlt_pack_begin.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    hip.jxlt_pack_sizes.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    hip.jxlt_pack_deliver.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.jxlt_output_buffer.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    sizes = built.PackedSections()
    out = C.c_void_p()
    assert hip.jxlt_output_buffer(fresh._ctx, 1 << 20, C.byref(out)) == 0
    assert hip.jxlt_histograms_ready(fresh._ctx) < 0              # nothing enqueued
    assert hip.jxlt_pack_begin(fresh._ctx, 1, table.ctypes.data) < 0
    assert hip.jxlt_pack_sizes(fresh._ctx, 1, C.byref(sizes)) < 0    # nothing measured
    assert hip.jxlt_pack_deliver(fresh._ctx, 1, out, None, 0, 0) < 0  # nothing measured
    assert hip.jxlt_pack_begin(fresh._ctx, 2, table.ctypes.data) < 0  # no such kind
    fresh.upload(planes)
    fresh.enqueue(1.0, 0)
    assert hip.jxlt_pack_begin(fresh._ctx, 1, table.ctypes.data) < 0   # AC sections need the histograms first
    fresh.synchronize()
    ac, dc = fresh.fetch_histograms()
    assert hip.jxlt_histograms_ready(fresh._ctx) == 1
    at, dt = built.build_code_tables(ac, dc)
    assert hip.jxlt_pack_begin(fresh._ctx, 0, dt.ctypes.data) == 0
    run = (C.c_uint32 * 4)(0, 5, 0, 0)  # (first 0, five sections of a one-section frame)
    assert hip.jxlt_pack_deliver(fresh._ctx, 0, out, run, 1, 0) < 0    # a run beyond the sections
    assert hip.jxlt_pack_deliver(fresh._ctx, 0, out, run, 1, 1) < 0    # runs are not end-aligned
    assert hip.jxlt_pack_deliver(fresh._ctx, 0, out, None, 0, 0) == 0
    fresh.synchronize()
    # The sections travel by copy commands since round 5, so ANY destination a copy command takes is accepted
    # (until then a kernel stored the bytes and pageable memory was refused): ordinary host memory is the
    # synchronous fallback and receives the same bytes as the context's page-locked buffer.
    assert hip.jxlt_pack_sizes(fresh._ctx, 0, C.byref(sizes)) == 0
    nbytes = int(sizes.section_offset[sizes.num_sections])
    assert 0 < nbytes < (1 << 16)
    pinned_bytes = C.string_at(out, nbytes)
    pageable = np.full(1 << 16, 0xAB, np.uint8)
    assert hip.jxlt_pack_deliver(fresh._ctx, 0, pageable.ctypes.data, None, 0, 0) == 0
    fresh.synchronize()
    assert pageable[:nbytes].tobytes() == pinned_bytes and (pageable[nbytes + 8:] == 0xAB).all()
    fresh.close()


def test_values_outside_unit_range(enc):
    rng = np.random.default_rng(7)
    planes = (rng.random((3, 72, 136)) * 3.0 - 1.0).astype(np.float32)  # [-1, 2): negative + HDR
    want = T.oracle_hot_path(planes, 1.0)
    got = enc.hot_path(planes, 1.0, debug=True)
    assert T.compare_results(want, got, "oracle", "gpu") == []


def test_small_frames_give_the_same_tokens_every_time(enc):
    """The nonzero masks leave tile_kernel by scalar stores through the scalar data cache (jxlt_device_common.h:
    JXLT_SCALAR_STORE64 / JXLT_SCALAR_STORES_DONE).  On a frame of a few tiles nothing but the kernel's own
    write-back puts them in memory before token_kernel reads them: with the write-back issued ahead of the stores it
    was meant to cover (scalar memory operations complete out of order) one context byte in this frame's 26 578
    records was wrong in about every third run (round 5).  Sixty runs here."""
    rng = np.random.default_rng(7)
    planes = (rng.random((3, 72, 136)) * 3.0 - 1.0).astype(np.float32)  # the frame of test_values_outside_unit_range
    want = T.oracle_hot_path(planes, 1.0)
    wrong = []
    for i in range(60):
        got = enc.hot_path(planes, 1.0)
        if got.all_tokens() != want.all_tokens():
            wrong.append(i)
    assert wrong == []


def test_full_size_properties(built, enc):
    """BASELINE-scale input (4096x4096 = config #2 size, too slow for a full oracle
    pass in the test budget): size-independent properties instead.
      * determinism: two passes give identical bytes
      * shard independence: any group-aligned crop encodes to the same per-group
        token streams and side-band cells as those groups of the full frame
      * the crop itself is checked against the oracle."""
    size = 4096
    img = T.synthetic_image(size, size)
    planes = T.to_planes(img)
    a = enc.hot_path(planes, 1.0)
    b = enc.hot_path(planes, 1.0)
    assert a.all_tokens() == b.all_tokens()
    assert np.array_equal(a.quant_dc, b.quant_dc)
    gpr = size // 256
    x0, y0, s = 2048, 1024, 768  # a 3x3-group window away from the origin
    crop = np.ascontiguousarray(planes[:, y0:y0 + s, x0:x0 + s])
    c = enc.hot_path(crop, 1.0)
    for gy in range(s // 256):
        for gx in range(s // 256):
            full_g = (y0 // 256 + gy) * gpr + (x0 // 256 + gx)
            assert c.group_tokens[gy * (s // 256) + gx] == a.group_tokens[full_g], (gx, gy)
    bs = slice(y0 // 8, (y0 + s) // 8), slice(x0 // 8, (x0 + s) // 8)
    assert np.array_equal(c.strategy, a.strategy[bs])
    assert np.array_equal(c.raw_quant, a.raw_quant[bs])
    assert np.array_equal(c.quant_dc, a.quant_dc[(slice(None),) + bs])
    want = T.oracle_hot_path(crop, 1.0)
    assert T.compare_results(want, c, "oracle", "gpu", check_debug=False) == []


def test_device_resident_entry_point(built, enc):
    """jxlt_image_set_device: planes living in a torch tensor (pitch != width)."""
    import torch
    w, h = 200, 137
    planes = T.to_planes(T.synthetic_image(w, h))
    t = torch.zeros((3, h, 256), dtype=torch.float32, device="cuda")
    t[:, :, :w] = torch.from_numpy(planes).cuda()
    torch.cuda.synchronize()
    enc.set_device_image([t[c].data_ptr() for c in range(3)], 256 * 4, w, h, keepalive=t)
    jxl = enc.encode_resident(1.0)
    assert jxl == T.assemble_codestream(T.oracle_hot_path(planes, 1.0), 1.0)


@pytest.mark.parametrize("w,h,distance", [(700, 520, 1.0), (2100, 300, 2.0), (256, 256, 1.0), (1030, 1030, 0.5)])
def test_device_packed_sections_equal_host_packed(built, enc, w, h, distance):
    """Production route (symbol histograms + section bit packing on the device) vs the
    raw-token route (tokens copied to the host, packed by the host back-end) vs the oracle."""
    planes = T.to_planes(T.synthetic_image(w, h, hard=(w == 700)))
    enc.upload(planes)
    a = enc.encode_resident(distance)
    b = enc.encode_resident_raw_tokens(distance)
    assert a == b
    assert enc.encode_resident(distance, copy=False).tobytes() == a  # page-locked view variant
    assert a == T.assemble_codestream(T.oracle_hot_path(planes, distance), distance)


def test_output_buffer_keeps_early_sections_when_it_grows(built):
    """The DC-group sections leave for the context's page-locked output buffer before the AC sections' size is
    known (jxlt_pack_deliver, end-aligned); when the frame turns out larger than the buffer (sized by the previous
    frame), jxlt_output_buffer grows WITH its contents.  A fresh context, frames of
    rising, falling and rising size, each several times (the second encode of a size finds the buffer right)."""
    e = built.Encoder(0)
    try:
        for w, h, d, hard in [(64, 64, 1.0, False), (2100, 1300, 0.5, True), (300, 200, 2.0, False),
                              (2600, 2100, 1.0, False), (520, 264, 1.0, True)]:
            planes = T.to_planes(T.synthetic_image(w, h, seed=w + h, hard=hard))
            want = T.assemble_codestream(T.oracle_hot_path(planes, d), d)
            e.upload(planes)
            for _ in range(3):
                assert e.encode_resident(d, copy=False).tobytes() == want, (w, h)
    finally:
        e.close()


def test_multi_encoder_two_contexts_equal_single_device(built):
    """jxlt_multi_encoder_* (BASELINE config #4 behind the C boundary): ONE frame cut into row slabs of whole
    DC groups over two device contexts (both on GPU 0 here), histograms summed on the host, every context
    writing its sections into the one output buffer -- must equal the single-context codestream and the
    oracle's, from host planes (pageable and page-locked: the pipelined upload), from a raw PFM payload and
    from slabs that already are in device memory."""
    import torch
    w, h, d = 520, 2048 + 2048 + 300, 1.0   # three DC-group rows: slabs of 2 + 1
    planes = T.to_planes(T.synthetic_image(w, h))
    want = T.assemble_codestream(T.oracle_hot_path(planes, d), d)
    me = built.MultiEncoder([0, 0])
    assert me.encode(planes, d).tobytes() == want
    pinned, owner = built.pinned_empty((3, h, w + 8))
    pinned[:, :, :w] = planes
    assert me.encode(pinned[:, :, :w], d).tobytes() == want
    assert me.encode(planes, d).tobytes() == want  # reusable
    payload = T.pfm_payload(planes, big_endian=True)
    assert me.encode_pfm(payload.view(np.uint8), w, h, True, d).tobytes() == want
    # slabs resident in device memory
    t = torch.from_numpy(planes).cuda()
    for slab in range(2):
        x0, y0, x1, y1 = built.shard_rect(w, h, 2, slab)
        me.set_device_slab(slab, [t[c, y0:, x0:].data_ptr() for c in range(3)], w * 4, x1 - x0, y1 - y0, keepalive=t)
    assert me.encode_resident(w, h, d).tobytes() == want
    # a frame of a single DC-group row takes the ordinary path on the first context
    small = T.to_planes(T.synthetic_image(300, 264))
    assert me.encode(small, 2.0).tobytes() == T.assemble_codestream(T.oracle_hot_path(small, 2.0), 2.0)
    assert built.encode_file(planes, d) == want
    me.close()
    # three contexts, one of them without rows (two DC-group rows)
    w2, h2 = 300, 2048 + 100
    p2 = T.to_planes(T.synthetic_image(w2, h2, seed=5))
    me3 = built.MultiEncoder([0, 0, 0])
    assert me3.encode(p2, 0.5).tobytes() == T.assemble_codestream(T.oracle_hot_path(p2, 0.5), 0.5)
    me3.close()


def test_multi_encoder_shards_dc_groups_by_index(built):
    """Round 4: a frame's DC groups are dealt out as rectangles, not as rows only.  A frame of
    ONE row of DC groups over four contexts (until round 3: "nothing to shard"), 8192 x 8192 = 4 x 4 DC groups over
    eight contexts (2 x 1 each; until round 3 at most four could work), an uneven 3 x 2 over five -- from host planes
    and from rectangles that already are in device memory; every codestream equals the single-context one, which the
    other tests pin to the oracle."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    enc1 = built.Encoder(0)
    for w, h, n, d in [(4 * 2048 - 100, 300, 4, 1.0), (2 * 2048 + 300, 2048 + 20, 5, 2.0), (8192, 8192, 8, 1.0)]:
        t = bench.frame_rows_on_device(torch, w, 0, (h + 1023) // 1024 * 1024, 7, dev)[:, :h].contiguous()
        torch.cuda.synchronize()
        enc1.set_device_image([t[c].data_ptr() for c in range(3)], w * 4, w, h, keepalive=t)
        single = enc1.encode_resident(d)
        me = built.MultiEncoder([0] * n)
        rects = [built.shard_rect(w, h, n, r) for r in range(n)]
        assert sum(1 for r in rects if r[2] > r[0]) == min(n, ((w + 2047) // 2048) * ((h + 2047) // 2048))
        for slab, (x0, y0, x1, y1) in enumerate(rects):
            if x1 > x0:
                me.set_device_slab(slab, [t[c, y0:, x0:].data_ptr() for c in range(3)], w * 4, x1 - x0, y1 - y0, keepalive=t)
        assert me.encode_resident(w, h, d).tobytes() == single, (w, h, n)
        if w * h < (1 << 24):  # ... and from host planes, each context uploading its own rectangle
            assert me.encode(t.cpu().numpy(), d).tobytes() == single, (w, h, n)
        me.close()
        del t
    enc1.close()


def test_two_threads_with_two_device_lists_encode_side_by_side(built):
    """The device list of jxl::SetEncoderDevices and the encoder built on it belong to the calling thread (round 5;
    until round 4: one process-wide encoder behind a mutex, concurrent callers serialised -- VERDICT r4 item 7a).  Two
    threads with two lists -- two and three contexts on the one GPU -- encode frames of several DC groups at the same
    time, three frames each; every codestream is the oracle's; a third thread that never named a list takes the
    single-device path meanwhile."""
    import threading
    frames = {"a": (300, 4300, 1.0, 21), "b": (4300, 300, 2.0, 22), "c": (700, 520, 1.0, 23)}
    planes = {k: T.to_planes(T.synthetic_image(w, h, seed=s)) for k, (w, h, d, s) in frames.items()}
    want = {k: bytes(T.oracle_encode_file(planes[k], frames[k][2], nthreads=8)[0]) for k in frames}
    errors, started = [], threading.Barrier(3)

    def worker(key, devices):
        try:
            started.wait(timeout=60)
            for _ in range(3):
                got = built.encode_file_devices(planes[key], frames[key][2], devices) if devices else \
                    built.encode_file(planes[key], frames[key][2])
                if got != want[key]:
                    errors.append("%s over %s: bytes differ" % (key, devices))
        except Exception as e:  # noqa: BLE001 (reported by the main thread)
            errors.append("%s over %s: %r" % (key, devices, e))

    threads = [threading.Thread(target=worker, args=("a", [0, 0])), threading.Thread(target=worker, args=("b", [0, 0, 0])),
               threading.Thread(target=worker, args=("c", None))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads)


def test_frames_of_one_column_or_one_row_of_groups(built):
    """Narrow / tall and wide / flat frames: hundreds of groups and dozens of DC groups in ONE column or row (every
    section is small, the sections are many -- the look-back of the single pass crosses a section start in almost
    every tile, the tile plan is all first tiles), a frame eight pixels wide, one nine pixels high.  Against the
    oracle, twice in a row on one context."""
    for (w, h, d) in [(256, 65536, 1.0), (65536, 200, 2.0), (300, 30000, 0.5), (8, 40000, 1.0), (50000, 9, 1.0)]:
        img = T.to_planes(T.synthetic_image(w, h, seed=w ^ h))
        want = bytes(T.oracle_encode_file(img, d, nthreads=8)[0])
        e = built.Encoder(0)
        e.upload(img)
        assert bytes(e.encode_resident(d)) == want, (w, h)
        assert bytes(e.encode_resident(d)) == want, (w, h)
        e.close()


@pytest.mark.parametrize("w,h,d", [(2049, 2049, 0.7), (2056, 2049, 1.0), (4097, 2049, 2.0), (8200, 4100, 1.0)])
def test_frames_whose_corner_dc_group_is_one_block(built, w, h, d):
    """Frames that end 1..8 pixels behind a DC-group boundary on both axes: the corner DC group is ONE block (or a
    row / column of blocks one block thick).  The reference traps on them (enc_frame.cc:335-339 -> the CfL allotment of
    a 1 x 1 DC group, base/padded_bytes.h:174) -- the product encodes them like any other frame: the
    oracle's bytes, and the first one is read back by the independent decoder."""
    planes = T.to_planes(T.synthetic_image(w, h, seed=216 if w == 8200 else 102))  # 8200 x 4100: VERDICT r5's 18th frame
    want = bytes(T.oracle_encode_file(planes, d, nthreads=8)[0])
    assert built.encode_file(planes, d) == want
    e = built.Encoder(0)
    e.upload(planes)
    assert bytes(e.encode_resident(d)) == want
    e.close()
    if (w, h) == (2049, 2049):
        import jxl_decoder as D
        dec = D.decode(want)
        assert (dec.xsize, dec.ysize) == (w, h)
        assert D.psnr_opsin_db(planes, dec.linear_rgb) > 37.0


@pytest.mark.parametrize("poison", [1e38, float("inf"), float("nan"), -1e38], ids=["1e38", "inf", "nan", "-1e38"])
def test_values_the_format_cannot_carry_are_refused(built, poison):
    """40 samples of 1e38 / +Inf / NaN / -1e38 in a 136 x 72 frame.  A quantised coefficient
    whose token does not fit 16 bits makes the reference trap in debug builds (enc_bit_writer.cc:120) and write a
    stream no decoder accepts otherwise; the device counts such tiles and every entry point answers
    JXLT_ERR_UNSUPPORTED (-4) -- or the stream is one the independent reader accepts to the last bit.  The context
    encodes ordinary frames afterwards as if nothing had happened."""
    import jxl_decoder as D
    img = T.synthetic_image(136, 72, seed=3)
    idx = np.random.default_rng(7).integers(0, img.size, size=40)
    img.reshape(-1)[idx] = np.float32(poison)
    planes = T.to_planes(img)
    e = built.Encoder(0)
    e.upload(planes)
    refused = False
    try:
        jxl = bytes(e.encode_resident(1.0))
    except built.JxlTinyError as err:
        refused = True
        assert "(-4)" in str(err), str(err)
    if poison > 1e30:
        assert refused
    if not refused:
        D.decode(jxl)
    # ... and the raw-token route says the same
    try:
        e.hot_path(planes, 1.0)
        assert not refused
    except built.JxlTinyError as err:
        assert refused and "(-4)" in str(err), str(err)
    good = T.to_planes(T.synthetic_image(136, 72, seed=3))
    e.upload(good)
    assert bytes(e.encode_resident(1.0)) == T.assemble_codestream(T.oracle_hot_path(good, 1.0), 1.0)
    e.close()
    # the drop-in's entry point: false / an error, no bytes
    if refused:
        with pytest.raises(built.JxlTinyError):
            built.encode_file(planes, 1.0)


def test_single_pass_packing_of_several_contexts_on_one_device_does_not_stall(built):
    """Tiles of the single pass wait for the tiles in front of them.  Handed out by workgroup index, a tile could wait
    for one that was never dispatched because another context's waiting tiles held every slot of its XCD -- and the
    other way round: two contexts packing halves of a 16384^2 frame on one GPU took 12 s per frame (round 4, first
    version).  Tiles are handed out by ticket since (in the order in which workgroups START), so what a tile waits
    for is running or done whatever shares the device.  Here: the single pass forced for every size, two and four
    contexts on one GPU, launches of thousands of tiles each, in a child process with a time limit."""
    import os
    import sys
    script = """
import sys, time
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
import __graft_entry__ as G
import bench
pkg = G.load_package()
size = 8192
t = bench.frame_rows_on_device(torch, size, 0, size, 3, torch.device("cuda", 0))
torch.cuda.synchronize()
enc = pkg.Encoder(0)
enc.set_device_image([t[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=t)
single = bytes(enc.encode_resident(1.0))
worst = 0.0
for n in (2, 4):
    me = pkg.MultiEncoder([0] * n)
    for slab in range(n):
        x0, y0, x1, y1 = pkg.shard_rect(size, size, n, slab)
        me.set_device_slab(slab, [t[c, y0:, x0:].data_ptr() for c in range(3)], size * 4, x1 - x0, y1 - y0, keepalive=t)
    for rep in range(6):
        t0 = time.perf_counter()
        out = me.encode_resident(size, size, 1.0).tobytes()
        worst = max(worst, time.perf_counter() - t0) if rep else worst
        assert out == single, (n, rep)
    me.close()
print("RESULT %%.4f" %% worst)
""" % (str(T.ROOT), str(T.ROOT / "tests"))
    env = dict(os.environ, JXLT_PACK_TWO_PASS="0")
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    assert line, r.stdout + r.stderr
    assert float(line[0].split()[1]) < 0.1, line  # (a frame takes ~2 ms; the stall was seconds)


def test_multi_encoder_grows_its_output_region_and_reports_errors(built):
    """Incompressible content at a tiny distance needs more than the byte per pixel the output region starts
    with: the encoder redoes the frame with the packer's worst case.  Then: frames of changing geometry through
    one encoder, and argument errors."""
    rng = np.random.default_rng(11)
    w, h, d = 520, 2048 + 520, 0.03
    noise = rng.random((3, h, w)).astype(np.float32) * 4.0 - 1.5
    want = T.assemble_codestream(T.oracle_hot_path(noise, d), d)
    assert len(want) > w * h + (1 << 20)  # really beyond the initial capacity
    me = built.MultiEncoder([0, 0])
    assert me.encode(noise, d).tobytes() == want
    for ww, hh, dd in [(300, 2048 + 8, 1.0), (2100, 2048 + 2048 + 1, 3.0), (64, 4096, 0.5)]:
        p = T.to_planes(T.synthetic_image(ww, hh, seed=ww))
        assert me.encode(p, dd).tobytes() == T.assemble_codestream(T.oracle_hot_path(p, dd), dd), (ww, hh)
    with pytest.raises(built.JxlTinyError):
        me.encode(T.to_planes(T.synthetic_image(64, 4096)), 0.0)  # lossless is not supported (enc_file.cc:60-62)
    with pytest.raises(built.JxlTinyError):
        me.encode_resident(64, 4096, 1.0)  # no slabs were set for this geometry
    me.close()
    with pytest.raises(built.JxlTinyError):
        built.MultiEncoder([0, 99])  # no such device


def test_attached_frame_with_redone_tiles(built, enc):
    """A frame that arrives from host memory in pieces (rows of DC groups, the last row in rows of groups), every
    tile of which is redone with computed square roots (flag 0x1000 makes every tile of tile_kernel report a
    table overflow): each piece's launch has its own list of tiles and its own redo launch."""
    w, h, d = 300, 2048 + 264, 2.0
    planes = T.to_planes(T.synthetic_image(w, h))
    pinned, owner = built.pinned_empty((3, h, w))
    pinned[...] = planes
    enc.attach_host(pinned)
    dp = enc.enqueue(d, 0x1000)
    fr = enc.fetch_raw()
    st = enc.stats()
    assert st["tiles_redone_exact_roots"] == st["tiles"] == ((w + 63) // 64) * ((h + 63) // 64)
    frame = enc.assemble(fr, dp, 0)
    assert built.file_header(w, h) + frame == T.assemble_codestream(T.oracle_hot_path(planes, d), d)


def _shard_rank(rank, world, name, w, h, d, q, barrier):
    try:
        import sys
        sys.path.insert(0, str(T.ROOT / "tests"))
        pkg = T.product()
        planes = T.to_planes(T.synthetic_image(w, h))
        x0, y0, x1, y1 = pkg.shard_rect(w, h, world, rank)
        enc = pkg.Encoder(0)
        if y1 > y0:
            enc.upload(np.ascontiguousarray(planes[:, y0:y1, x0:x1]))
        grp = pkg.ShardGroup(name, 0, world, 8 << 20, 4096) if rank == 0 else None
        barrier.wait()
        if grp is None:
            grp = pkg.ShardGroup(name, rank, world, 8 << 20, 4096)
        out = []
        for _ in range(2):
            v = grp.encode(enc, w, h, d)
            out.append(v.tobytes() if v is not None else None)
        barrier.wait()
        grp.close()
        enc.close()
        q.put((rank, out))
    except Exception as e:  # pragma: no cover
        q.put((rank, "ERROR: %r" % (e,)))


def test_shard_group_two_processes_on_one_gpu(built):
    """jxlt_shard_group_* / jxlt_shard_encode: the one-process-per-GPU form bench.py's ranks use -- two
    processes (both on GPU 0 here), the output area in POSIX shared memory page-locked by each of them,
    each process's GPU copying its sections straight into it."""
    import multiprocessing as mp
    import os
    w, h, d = 600, 2048 + 520, 2.0
    ctx = mp.get_context("spawn")
    q, barrier = ctx.Queue(), ctx.Barrier(2)
    name = "/jxlt-gputest-%d" % os.getpid()
    procs = [ctx.Process(target=_shard_rank, args=(r, 2, name, w, h, d, q, barrier)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    planes = T.to_planes(T.synthetic_image(w, h))
    want = T.assemble_codestream(T.oracle_hot_path(planes, d), d)
    assert results[0] == [want, want], results[0] if isinstance(results[0], str) else "codestream differs"
    assert results[1] == [None, None], results[1]


def test_bench_self_launch_two_ranks_on_one_gpu(built):
    """`python bench.py --gpus 2` without a launcher (what the driver may run): the script starts its two
    ranks itself; here both on GPU 0 (JXLT_BENCH_ONE_DEVICE).  The sharded codestream must equal the
    single-GPU one (the bench's own parity gate) and the run must exit 0 with one JSON line."""
    import json
    import os
    import sys
    env = dict(os.environ, JXLT_BENCH_ONE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, str(T.ROOT / "bench.py"), "--gpus", "2", "--size", "4096", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["parity_gate"]["sharded_equals_single_gpu_codestream"] is True
    assert "4096x4096" in out["config"]["workload"]


def _check_sampled_groups(planes, got, distance, dct8, picks):
    """Groups (gy, gx) of a full-frame GPU result against the oracle run on the matching 256x256 crops
    (an AC group depends on nothing outside itself): tokens and side-band grids."""
    size_y, size_x = planes.shape[1], planes.shape[2]
    gpr = (size_x + 255) // 256
    for gy, gx in picks:
        crop = planes[:, gy * 256:(gy + 1) * 256, gx * 256:(gx + 1) * 256]
        if hasattr(crop, "cpu"):  # (a frame that stays on the device: only the crops come to the host)
            crop = crop.cpu().numpy()
        crop = np.ascontiguousarray(crop)
        want = T.oracle_hot_path(crop, distance, dct8)
        assert got.group_tokens[gy * gpr + gx] == want.group_tokens[0], (gy, gx)
        hb, wb = want.raw_quant.shape
        bs = (slice(gy * 32, gy * 32 + hb), slice(gx * 32, gx * 32 + wb))
        assert np.array_equal(got.quant_dc[(slice(None),) + bs], want.quant_dc), (gy, gx)
        assert np.array_equal(got.raw_quant[bs], want.raw_quant) and np.array_equal(got.strategy[bs], want.strategy)
        ts = (slice(gy * 4, gy * 4 + want.ytox.shape[0]), slice(gx * 4, gx * 4 + want.ytox.shape[1]))
        assert np.array_equal(got.ytox[ts], want.ytox) and np.array_equal(got.ytob[ts], want.ytob), (gy, gx)


def _lattice(n, k):
    pts = sorted(set([0, n - 1] + [int(round(i * (n - 1) / (k - 1.0))) for i in range(k)]))
    return [(gy, gx) for gy in pts for gx in pts]


def test_baseline_config2_4096_fixed_dct8(built, enc):
    """BASELINE config #2 at its stated size: 4096 x 4096, fixed DCT8 strategy.  The whole frame is encoded once;
    64 groups on an 8 x 8 lattice over the WHOLE frame (corners, last row / column, interior) are compared with
    the oracle on the matching crops."""
    planes = T.to_planes(T.synthetic_image(4096, 4096))
    got = enc.hot_path(planes, 1.0, force_dct8=True)
    assert (got.strategy == 1).all()
    _check_sampled_groups(planes, got, 1.0, True, _lattice(16, 8))


def test_baseline_config3_8192_full_search(built, enc):
    """BASELINE config #3 at its stated size: 8192 x 8192, full strategy search + adaptive quantisation;
    64 groups sampled across the whole frame against the oracle, and the production codestream (device-side
    histograms + packing) equals the raw-token route's."""
    import torch
    size = 8192
    import bench  # the benchmark's own frame generator (repo root is on sys.path, tests/conftest.py)
    t = bench.frame_rows_on_device(torch, size, 0, size, 0, torch.device("cuda", 0))
    planes = t.cpu().numpy()
    enc.set_device_image([t[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=t)
    enc.enqueue(1.0, 0)
    got = built.HotPathOutput(enc.fetch_raw())
    _check_sampled_groups(planes, got, 1.0, False, _lattice(32, 8))
    a = enc.encode_resident(1.0)
    assert a == enc.encode_resident_raw_tokens(1.0)
    # the same frame from page-locked host memory with the upload pipelined under the kernels
    pinned, owner = built.pinned_empty((3, size, size))
    pinned[...] = planes
    enc.attach_host(pinned)
    assert enc.encode_resident(1.0) == a
    payload, owner2 = built.pinned_empty((size * size * 3,))
    payload[...] = T.pfm_payload(planes)
    enc.attach_host_pfm(payload, size, size)
    assert enc.encode_resident(1.0) == a


BENCH_FRAME_SHA16 = "6d6354ebdcc0222b"  # bench.py's config.codestream_sha256 of its default frame (BENCH_r02.json)


def test_baseline_config4_16384_frame(built, enc):
    """BASELINE config #4 at its stated size: the benchmark's own 16384 x 16384 frame.  64 groups on an 8 x 8
    lattice over the WHOLE frame against the oracle on the matching crops; the codestream is the one the
    benchmark reports (sha-256 of the bytes in BENCH_r*.json); the same frame cut over two device contexts
    (jxlt_multi_encoder_*, both on GPU 0 here) and over two PROCESSES (bench.py --gpus 2, what the driver's
    scaling run launches; both ranks on GPU 0 here) gives the same bytes."""
    import hashlib
    import json
    import os
    import sys
    import torch
    import bench
    size = 16384
    t = bench.frame_rows_on_device(torch, size, 0, size, 0, torch.device("cuda", 0))
    enc.set_device_image([t[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=t)
    enc.enqueue(1.0, 0)
    got = built.HotPathOutput(enc.fetch_raw())
    planes = t.cpu().numpy()
    _check_sampled_groups(planes, got, 1.0, False, _lattice(64, 8))
    del planes
    single = enc.encode_resident(1.0)
    assert hashlib.sha256(single).hexdigest()[:16] == BENCH_FRAME_SHA16
    assert enc.stats()["tiles_redone_exact_roots"] == 0
    me = built.MultiEncoder([0, 0])
    for slab in range(2):
        x0, y0, x1, y1 = built.shard_rect(size, size, 2, slab)
        me.set_device_slab(slab, [t[c, y0:, x0:].data_ptr() for c in range(3)], size * 4, x1 - x0, y1 - y0, keepalive=t)
    assert me.encode_resident(size, size, 1.0).tobytes() == single
    me.close()
    del t
    torch.cuda.empty_cache()
    env = dict(os.environ, JXLT_BENCH_ONE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, str(T.ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 2 and "16384x16384" in out["config"]["workload"]
    assert out["parity_gate"]["sharded_equals_single_gpu_codestream"] is True
    assert out["config"]["codestream_sha256"] == BENCH_FRAME_SHA16


def test_frames_in_a_row_whose_ac_code_is_ready_first(built, enc):
    """Small frames: the AC histogram arrives before the DC code is built, so the host builds both codes at the same
    time and has the AC sections measured, written and on their way before the DC-group sections exist -- at a place
    that is fixed before any size is known (a bound of what stands in front: host/enc_frame.cc); the DC-group
    sections are set against them from the right by the device.  A row of frames of one size whose DC-group sections
    shrink and grow (flat, busy, flat ...): every codestream equals the oracle's."""
    w, h = 1096, 840
    busy = T.to_planes(T.synthetic_image(w, h, hard=True))
    calm = T.to_planes(T.synthetic_image(w, h))
    flat = np.full_like(calm, 0.25)
    flat[:, ::64, ::64] = 0.3
    want = {}
    for name, planes in (("flat", flat), ("calm", calm), ("busy", busy)):
        want[name] = T.assemble_codestream(T.oracle_hot_path(planes, 1.0), 1.0)
    assert len(want["busy"]) > 2 * len(want["flat"])
    frames = {"flat": flat, "calm": calm, "busy": busy}
    for name in ("flat", "flat", "busy", "busy", "calm", "flat", "busy", "calm", "calm"):
        enc.upload(frames[name])
        assert enc.encode_resident(1.0) == want[name], name
        assert enc.encode_resident(1.0, copy=False).tobytes() == want[name], name
    # a size at which the DC-group sections of an ordinary frame are hundreds of KB, many times a flat frame's
    size = 4096
    calm = T.to_planes(T.synthetic_image(size, size))
    flat = np.full_like(calm, 0.25)
    flat[:, ::64, ::64] = 0.3
    frames = {"flat": flat, "calm": calm}
    want = {k: T.assemble_codestream(T.oracle_hot_path(v, 1.0), 1.0) for k, v in frames.items()}
    dc_bytes = {}
    for name in ("flat", "flat", "calm", "calm", "flat", "calm"):
        enc.upload(frames[name])
        # (the codestream in the context's buffer: jxlt_encode_resident_view, the entry point the bench uses)
        assert enc.encode_resident(1.0, copy=False).tobytes() == want[name], name
        # (a second measuring pass behind a complete encode: until round 3 it ran on the used-up tile plan)
        sizes = enc.pack_sections(0, built.build_code_tables(*enc.fetch_histograms())[1])
        dc_bytes[name] = int(sizes[1][-1])
    assert dc_bytes["calm"] > dc_bytes["flat"] * 5 // 4 + (96 << 10), dc_bytes


@pytest.mark.parametrize("throughput_mode", [0, 1])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_sequences_of_calls_on_one_context(built, enc, seed, throughput_mode):
    """The C ABI's calls in random order on ONE context: frames of three sizes set in turn, complete encodes through
    both entry points and through the raw-token route, the device pipeline alone, section packing behind a complete
    encode (a second measuring pass), statistics in between.  Every codestream and every packed section must be what
    a fresh context gives -- state left behind by one call must not leak into the next.  throughput_mode: the
    context as a lane of a batch holds it (jxlt_context_set_wait_mode(ctx, 1)): the two kinds' section packing shares its
    launches, the DC-group sections' pass is put off until the AC sections' is asked for (or until somebody wants its sizes),
    and a frame's last publication leaves the counters clean for the next frame, whatever its size."""
    rng = np.random.default_rng(seed)
    enc.set_wait_mode(throughput_mode)
    frames = [T.to_planes(T.synthetic_image(w, h, seed=40 + i, hard=(i == 1)))
              for i, (w, h) in enumerate([(300, 264), (96, 72), (520, 2100)])]
    want = [T.assemble_codestream(T.oracle_hot_path(p, 1.0), 1.0) for p in frames]
    tokens = [T.oracle_hot_path(p, 1.0).all_tokens() for p in frames]
    cur = 0
    enc.upload(frames[cur])
    encoded = False
    for step in range(70):
        op = int(rng.integers(0, 8))
        if op == 0:
            cur = int(rng.integers(0, 3))
            enc.upload(frames[cur])
            encoded = False
        elif op == 1:
            assert enc.encode_resident(1.0) == want[cur], (step, cur)
            encoded = True
        elif op == 2:
            assert enc.encode_resident(1.0, copy=False).tobytes() == want[cur], (step, cur)
            encoded = True
        elif op == 3:
            assert enc.encode_resident_raw_tokens(1.0) == want[cur], (step, cur)
            encoded = True
        elif op == 4:
            enc.enqueue(1.0, 0)
            assert built.HotPathOutput(enc.fetch_raw()).all_tokens() == tokens[cur], (step, cur)
            encoded = True
        elif op == 5 and encoded:
            ac, dc = enc.fetch_histograms()
            at, dt = built.build_code_tables(ac, dc)
            kind = int(rng.integers(0, 2))
            data, off, bits = enc.pack_sections(kind, at if kind else dt)
            assert len(data) == int(off[-1]) and ((bits + 7) // 8 == np.diff(off)).all(), (step, cur, kind)
            if kind == 1 and cur != 1:  # the AC sections are the tail of the codestream (not of a single-group frame's,
                                        # whose sections are concatenated bit by bit)
                assert want[cur].endswith(data.tobytes()), (step, cur)
        elif op == 6 and encoded:
            assert enc.stats()["tiles_redone_exact_roots"] == 0
        elif op == 7 and encoded:
            if throughput_mode:  # (a batch lane's frames carry no stage events)
                with pytest.raises(built.JxlTinyError):
                    enc.kernel_times()
            else:
                assert set(enc.kernel_times()) == {"tile_kernel", "tokenisation_after_tile_kernel"}
    enc.set_wait_mode(0)  # (the context is the module's)


def test_frame_above_one_gigapixel(built, enc):
    """The reference takes frames of up to 2^30 - 1 pixels per side (enc_file.cc:41-43); the device path indexes
    blocks with 32 bits and coefficients with 64.  32768 x 45056 = 1.48 Gpixel = 23.1 M blocks: past the 2^24 blocks
    that were the limit until round 3, and past the 22.4 M blocks from which a coefficient index needs more than
    32 bits (the last five rows of groups lie beyond).  Groups over the whole frame against the oracle on the
    matching crops; the codestream of the device-side packing equals the one the host packs from the raw tokens
    and the one of the frame cut over two device contexts."""
    import torch
    import bench
    xs, ys = 32768, 45056
    assert (xs // 8) * (ys // 8) * 192 > 1 << 32
    t = bench.frame_rows_on_device(torch, xs, 0, ys, 7, torch.device("cuda", 0))
    enc.set_device_image([t[c].data_ptr() for c in range(3)], xs * 4, xs, ys, keepalive=t)
    enc.enqueue(1.0, 0)
    got = built.HotPathOutput(enc.fetch_raw())
    assert got.strategy.shape == (ys // 8, xs // 8)
    first_beyond = ((1 << 32) // 192) // (xs // 8) // 32  # row of groups in which the 32-bit index would wrap
    rows = [0, 60, first_beyond - 1, first_beyond, first_beyond + 1, ys // 256 - 1]
    _check_sampled_groups(t, got, 1.0, False, [(gy, gx) for gy in rows for gx in (0, 77, xs // 256 - 1)])
    del got
    single = enc.encode_resident(1.0)
    assert enc.stats()["tiles"] == (xs // 64) * (ys // 64)
    assert enc.encode_resident_raw_tokens(1.0) == single
    me = built.MultiEncoder([0, 0])
    for slab in range(2):
        x0, y0, x1, y1 = built.shard_rect(xs, ys, 2, slab)
        me.set_device_slab(slab, [t[c, y0:, x0:].data_ptr() for c in range(3)], xs * 4, x1 - x0, y1 - y0, keepalive=t)
    assert me.encode_resident(xs, ys, 1.0).tobytes() == single
    me.close()


def test_baseline_config5_batch_of_32_frames_over_a_device_list(built):
    """BASELINE config #5's shape: a batch of 3840 x 2160 frames in page-locked host memory through ONE frame queue
    whose lanes sit on a list of devices (GPU 0 twice here): 32 frames (four different ones, eight times each),
    every codestream compared with the oracle's."""
    import concurrent.futures as cf
    distinct = [T.to_planes(T.synthetic_image(3840, 2160, seed=500 + i)) for i in range(4)]
    with cf.ThreadPoolExecutor(4) as pool:  # (the oracle releases the GIL)
        want = list(pool.map(lambda p: T.assemble_codestream(T.oracle_hot_path(p, 1.0), 1.0), distinct))
    keep, frames = [], []
    for p in distinct:
        arr, owner = built.pinned_empty(p.shape)
        arr[...] = p
        keep.append(owner)
        frames.append(arr)
    enc = built.BatchEncoder(lanes=4, devices=[0, 0])
    got = enc.encode([frames[i % 4] for i in range(32)], 1.0)
    enc.close()
    assert len(got) == 32
    for i in range(32):
        assert got[i] == want[i % 4], "frame %d" % i


def test_device_memory_of_a_destroyed_context_serves_the_next_one(built, enc):
    """While the device has a living context (`enc`), jxlt_context_destroy keeps device blocks of 1 MB and more for the
    next context of the process (memory that went through hipFree and comes back from hipMalloc is slower on this
    stack); jxlt_release_cached_memory returns them to the runtime.  Contexts in a row give the same
    bytes, the memory held does not grow with them, and after the release torch sees the memory free again."""
    import torch
    built.release_cached_memory()
    planes = T.to_planes(T.synthetic_image(2048, 1536))
    want = T.assemble_codestream(T.oracle_hot_path(planes, 1.0), 1.0)
    e = built.Encoder(0)  # (what the runtime itself allocates with the first context and kernel is not the cache's)
    e.upload(planes)
    assert e.encode_resident(1.0) == want
    e.close()
    built.release_cached_memory()
    free0 = torch.cuda.mem_get_info(0)[0]
    held = []
    for _ in range(4):
        e = built.Encoder(0)
        e.upload(planes)
        assert e.encode_resident(1.0) == want
        e.close()
        held.append(free0 - torch.cuda.mem_get_info(0)[0])
    assert held[0] > (20 << 20), held           # the first context's buffers are still allocated ...
    # ... and serve the later ones (a block may serve a request up to a quarter smaller than itself, so the second
    # context can add a block or two; from then on nothing grows)
    assert held[1] <= held[0] * 3 // 2 and held[3] <= held[1] + (8 << 20), held
    released = built.release_cached_memory(0)
    assert released >= (20 << 20)
    # (what the runtime keeps of freed memory for itself is not ours to return)
    assert held[3] - (free0 - torch.cuda.mem_get_info(0)[0]) >= released * 9 // 10
    assert built.release_cached_memory() == 0


def test_every_form_of_the_packing_stage_gives_the_oracles_bytes(built):
    """The packing stage has two forms -- one pass, in which every tile takes its bit position from the tiles in front
    of it (frames of up to 1024 groups), and measure + write (larger frames) -- and packs the DC-group sections on
    the main stream or on one of their own (above 1024 groups).  The size rules keep half of those combinations away
    from frames the suite can afford an oracle run for: here every combination is forced (JXLT_PACK_TWO_PASS,
    JXLT_DC_PACK_STREAM) on the same frames, in child processes, and must give the oracle's file -- several
    distances, a frame with ragged edges, one with more than one DC group, repeated encodes on one context."""
    import os
    import sys
    script = """
import sys, hashlib
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import __graft_entry__ as G
import jxlt_testlib as T
pkg = G.load_package()
out = []
for (w, h, d) in [(1000, 700, 1.0), (2304, 2100, 2.5), (4200, 520, 0.6)]:
    planes = T.to_planes(T.synthetic_image(w, h, seed=w + h))
    e = pkg.Encoder(0); e.upload(planes)
    for rep in range(3):
        data = bytes(e.encode_resident(d))
        out.append(hashlib.sha256(data).hexdigest())
    e.close()
print("RESULT", " ".join(out))
""" % (str(T.ROOT), str(T.ROOT / "tests"))
    want = []
    for (w, h, d) in [(1000, 700, 1.0), (2304, 2100, 2.5), (4200, 520, 0.6)]:
        import hashlib
        ref = T.oracle_encode_file(T.to_planes(T.synthetic_image(w, h, seed=w + h)), d)[0]
        want += [hashlib.sha256(bytes(ref)).hexdigest()] * 3
    for two_pass in ("0", "1"):
        for dc_stream in ("0", "1"):
            env = dict(os.environ, JXLT_PACK_TWO_PASS=two_pass, JXLT_DC_PACK_STREAM=dc_stream)
            r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=900)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
            assert line, r.stdout + r.stderr
            assert line[0].split()[1:] == want, (two_pass, dc_stream)


def test_nothing_is_kept_beyond_the_last_context_unless_asked_for(built):
    """Default: when the LAST context of a device is destroyed, the blocks kept for its successors go back to the
    runtime (a co-resident allocator sees the memory again without anybody calling jxlt_release_cached_memory, ADVICE
    r3); JXLT_DEVICE_CACHE_MB=<n> opts in to a cache that outlives the contexts.  In child processes: this one has
    living contexts of other tests."""
    import os
    import sys
    script = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import __graft_entry__ as G
import jxlt_testlib as T
pkg = G.load_package()
torch.cuda.init(); torch.zeros(1, device="cuda")
planes = T.to_planes(T.synthetic_image(2048, 1536))
keeper = pkg.Encoder(0)           # (the runtime's own first-use allocations happen here)
keeper.upload(planes); keeper.encode_resident(1.0)
free_with_keeper = torch.cuda.mem_get_info(0)[0]
e = pkg.Encoder(0); e.upload(planes); a = e.encode_resident(1.0); e.close()
held_beside_keeper = free_with_keeper - torch.cuda.mem_get_info(0)[0]
keeper.close()                    # the last context of the device
kept_after_last = pkg.release_cached_memory(0)
print("RESULT", held_beside_keeper, kept_after_last, len(a))
""" % (str(T.ROOT), str(T.ROOT / "tests"))
    def run(env_extra):
        env = dict(os.environ)
        env.pop("JXLT_DEVICE_CACHE_MB", None)
        env.update(env_extra)
        out = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        assert line, out.stdout + out.stderr
        return [int(v) for v in line[0].split()[1:]]
    held, kept, n = run({})
    assert held > (20 << 20) and kept == 0 and n > 0, (held, kept)       # kept beside a living context, gone with the last
    held, kept, n = run({"JXLT_DEVICE_CACHE_MB": "4096"})
    assert held > (20 << 20) and kept > (40 << 20), (held, kept)          # opted in: both contexts' blocks outlive them
    held, kept, n = run({"JXLT_DEVICE_CACHE_MB": "0"})
    assert kept == 0, (held, kept)   # 0: nothing is ever kept (what the runtime holds back of freed memory is its own)


def test_cached_device_memory_gives_way_when_memory_runs_out(built, enc):
    """What a destroyed context left for its successors must not stand in the way of a living one: a context whose
    allocation does not fit beside the cached blocks gets them released and tries again.  (A frame of half the size
    after a large one: its requests are too small for the cached blocks -- a block serves requests down to a
    quarter below its size -- and need memory of their own.)"""
    import torch
    import bench
    built.release_cached_memory()
    dev = torch.device("cuda", 0)
    large = bench.frame_rows_on_device(torch, 8192, 0, 8192, 3, dev)   # 8192 x 8192
    small = bench.frame_rows_on_device(torch, 8192, 0, 4096, 3, dev)   # 8192 x 4096: half the memory
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info(0)[0]
    e = built.Encoder(0)
    e.set_device_image([large[c].data_ptr() for c in range(3)], 8192 * 4, 8192, 8192, keepalive=large)
    a = e.encode_resident(1.0)
    e.close()
    held = free0 - torch.cuda.mem_get_info(0)[0]
    assert held > (400 << 20), held
    # leave a quarter of `held` free: the small frame's buffers (about half of `held`) do not fit -- unless the
    # cached blocks go
    filler = [torch.empty(torch.cuda.mem_get_info(0)[0] - held // 4, dtype=torch.uint8, device=dev)]
    # (the runtime may hold memory back that earlier tests freed and that mem_get_info does not report as free: take
    # that too, until a third of `held` can no longer be had)
    for _ in range(64):
        try:
            filler.append(torch.empty(held // 3, dtype=torch.uint8, device=dev))
        except torch.cuda.OutOfMemoryError:
            break
    e = built.Encoder(0)
    try:
        e.set_device_image([small[c].data_ptr() for c in range(3)], 8192 * 4, 8192, 4096, keepalive=small)
        b = e.encode_resident(1.0)
        assert 0 < len(b) < len(a)
        assert built.release_cached_memory(0) == 0  # (the first context's blocks went when the memory ran out)
    finally:
        # (whatever happens: the rest of the session must not run in the sliver of memory this test leaves)
        filler.clear()
        torch.cuda.empty_cache()
    # ... and the bytes are what a context in plenty of memory gives
    e2 = built.Encoder(0)
    e2.set_device_image([small[c].data_ptr() for c in range(3)], 8192 * 4, 8192, 4096, keepalive=small)
    assert e2.encode_resident(1.0) == b
    e.close()
    e2.close()
    built.release_cached_memory()


def test_thread_binding_next_to_the_device(built):
    """jxlt_bind_thread_near_device: the calling thread ends up on the CPUs the device's PCI function lists as
    local (or the call says the system does not tell); jxlt_device_count / jxlt_context_device agree with torch."""
    import os
    import threading
    import torch
    hip = built.hip_lib()
    hip.jxlt_device_count.restype = C.c_int
    hip.jxlt_bind_thread_near_device.argtypes = [C.c_int]
    hip.jxlt_context_device.argtypes = [C.c_void_p]
    assert hip.jxlt_device_count() == torch.cuda.device_count() >= 1
    enc = built.Encoder(0)
    assert hip.jxlt_context_device(enc._ctx) == 0
    enc.close()
    result = {}

    def run():  # (in a thread of its own: the test process keeps its affinity)
        before = os.sched_getaffinity(0)
        rc = hip.jxlt_bind_thread_near_device(0)
        after = os.sched_getaffinity(0)
        result.update(rc=rc, before=before, after=after)

    t = threading.Thread(target=run)
    t.start()
    t.join()
    assert result["rc"] in (0, -4), result  # JXLT_OK or JXLT_ERR_UNSUPPORTED
    assert len(result["after"]) >= 1
    if result["rc"] != 0:
        assert result["after"] == result["before"]


def test_attach_host_small_frames(built, enc):
    """jxlt_image_attach_host*: frames of one and of several DC-group rows, odd sizes, pitch != width,
    big-endian payload; debug intermediates through the slab-wise launches; refusal of pageable memory."""
    for w, h, d in [(200, 137, 1.0), (300, 2048 + 77, 2.0), (2100, 4100, 4.0)]:
        planes = T.to_planes(T.synthetic_image(w, h, seed=w))
        want = T.oracle_hot_path(planes, d)
        pinned, owner = built.pinned_empty((3, h, w + 16))
        pinned[:, :, :w] = planes
        enc.attach_host(pinned[:, :, :w])
        assert enc.encode_resident(d) == T.assemble_codestream(want, d), (w, h)
        payload, owner2 = built.pinned_empty((w * h * 3,))
        payload[...] = T.pfm_payload(planes, big_endian=True)
        enc.attach_host_pfm(payload, w, h, big_endian=True)
        enc.enqueue(d, built.FLAG_DEBUG_DUMP)
        fr = enc.fetch_raw()
        yb, xb = fr.ysize_blocks, fr.xsize_blocks

        def get(what, shape):
            a = np.empty(shape, np.float32)
            enc._check(enc._L.jxlt_debug_fetch(enc._ctx, what, a.ctypes.data, a.nbytes), "jxlt_debug_fetch")
            return a
        dbg = (np.stack([get(c, (yb * 8, xb * 8)) for c in range(3)]), get(3, (yb, xb)), get(4, (yb, xb)),
               get(5, (yb // 2 + 1, xb // 2 + 1, 8)))
        assert T.compare_results(want, built.HotPathOutput(fr, dbg), "oracle", "gpu") == [], (w, h)
    with pytest.raises(built.JxlTinyError, match="page-locked"):
        enc.attach_host(T.to_planes(T.synthetic_image(64, 64)))


def test_context_strategy_distance_survives_host_entry_points(built, enc):
    """jxlt_set_strategy_distance is a persistent per-context setting (include/jxl_tiny_amd.h): the host-level
    entry points must not reset it while the process-wide emulation flag is off."""
    planes = T.to_planes(T.synthetic_image(200, 137))
    T.set_strategy_distance(16.0)
    try:
        want = T.assemble_codestream(T.oracle_hot_path(planes, 0.5), 0.5)
    finally:
        T.set_strategy_distance(0.0)
    plain = T.assemble_codestream(T.oracle_hot_path(planes, 0.5), 0.5)
    assert want != plain
    enc.upload(planes)
    enc.set_strategy_distance(16.0)
    try:
        assert enc.encode_resident(0.5) == want
        assert enc.encode_resident(0.5, copy=False).tobytes() == want
    finally:
        enc.set_strategy_distance(0.0)
    assert enc.encode_resident(0.5) == plain


def test_hardware_shortcuts_are_exact_on_this_gpu():
    """tile_kernel replaces two generic IEEE sequences by shorter ones whose exactness depends on
    the accuracy of this GPU's v_rcp_f32 (jxlt_device_common.h: rcp_int_exact).  tools/rcp_probe checks
    the shortcut against IEEE division for every integer-valued float the quantiser can produce
    (all 2^32 - 1 non-zero int32 values); variant 2 is the one the kernel uses."""
    import pathlib
    root = pathlib.Path(__file__).resolve().parent.parent
    exe = root / "tools" / "rcp_probe"
    if not exe.exists():
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-o", str(exe),
                        str(root / "tools" / "rcp_probe.hip")], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=300).stdout
    lines = [l for l in out.splitlines() if l.startswith("variant 2 ")]
    assert len(lines) == 3, out
    for l in lines:
        assert "mismatches=0 " in l, l


def test_short_division_equals_ieee_on_this_gpu():
    """The kernels divide with v_rcp_f32 + the refinement steps of the IEEE expansion, without its operand
    scaling and fix-up (jxlt_device_common.h: div_normal).  tools/div_probe compares that with the compiler's
    division on 2^32 operand pairs (magnitudes 2^-40 .. 2^40) and on reciprocals."""
    import pathlib
    root = pathlib.Path(__file__).resolve().parent.parent
    exe = root / "tools" / "div_probe"
    if not exe.exists():
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-o", str(exe),
                        str(root / "tools" / "div_probe.hip")], check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "mismatches=0 of 4294967296; reciprocals: mismatches=0" in res.stdout, res.stdout


def test_square_root_by_rsq_is_exact_on_this_gpu():
    """The adaptive quantisation's MaskingSqrt takes its root from v_rsq_f32 + seven multiply-adds
    (jxlt_device_common.h: sqrt_exact_by_rsq).  tools/sqrt_rsq_probe compares that with IEEE sqrtf on every float
    between 2^-27 and 2^63 (755 M patterns)."""
    import pathlib
    root = pathlib.Path(__file__).resolve().parent.parent
    exe = root / "tools" / "sqrt_rsq_probe"
    if not exe.exists():
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-o", str(exe),
                        str(root / "tools" / "sqrt_rsq_probe.hip")], check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "rsq sequence: mismatches=0 of 754974720" in res.stdout, res.stdout


def test_lds_store_load_order_on_this_gpu():
    """tile_kernel's octet transposes store to LDS and load what OTHER lanes of the wave stored, with
    no s_waitcnt and no barrier in between: that relies on a wave's LDS operations executing in program
    order.  tools/lds_order_probe checks it for the layout in use (and a second one), 8 M elements each."""
    import pathlib
    root = pathlib.Path(__file__).resolve().parent.parent
    exe = root / "tools" / "lds_order_probe"
    if not exe.exists():
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-o", str(exe),
                        str(root / "tools" / "lds_order_probe.hip")], check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout
    assert res.stdout.count(": 0 wrong of") == 4, res.stdout


def test_addtid_lds_stores_honour_a_base_above_64_kb_on_this_gpu():
    """tile12_kernel's half transposes store rows through `s_mov_b32 m0, base; ds_write_addtid_b32`, and the
    fourth pair wave's rows lie at LDS byte offset 78 592 -- above the 16 bits of M0 that the gfx9 ISA text names as the
    instruction's base.  tools/lds_addtid_probe stores through bases on both sides of 64 KB (with and without an
    immediate offset) and reads all 96 KB back: every store where base + offset + 4 * lane says, nothing anywhere else."""
    import pathlib
    root = pathlib.Path(__file__).resolve().parent.parent
    exe = root / "tools" / "lds_addtid_probe"
    if not exe.exists():
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-o", str(exe),
                        str(root / "tools" / "lds_addtid_probe.hip")], check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout
    assert res.stdout.count(": 0 wrong of") == 16, res.stdout


def test_denormals_are_kept_by_the_device_code(built, enc):
    """the root-table offset of the entropy estimates is the bit pattern of the DENORMAL q x 2^-147
    (jxlt_tile_kernel.h); flushed to zero every offset would be 0 and nothing would notice.  The build pins
    -fno-gpu-flush-denormals-to-zero, and the context checks the product on the device when it is made
    (jxlt_context_create fails otherwise); here the same check through the testing header."""
    assert built.denormal_self_check(enc) == 12  # bits of 3.0f * 2^-147


def test_pfm_file_ingest_on_device(built, tmp_path):
    """jxlt_encode_pfm_file: a frame wider and taller than one group, big-endian payload, rows
    bottom-up -- read in place by tile_kernel."""
    img = T.synthetic_image(700, 520)
    pfm = tmp_path / "in.pfm"
    T.write_pfm(pfm, img, big_endian=True)
    got = built.encode_pfm_file(pfm, 2.0)
    want = T.assemble_codestream(T.oracle_hot_path(T.to_planes(img), 2.0), 2.0)
    assert got == want


def test_root_table_overflow_redo_on_gpu(built, enc):
    """Tiles in which tile_kernel meets a quantised magnitude beyond its square-root table are redone with
    computed roots by a second launch (flag 0x1000 makes every tile report one); jxlt_encode_stats counts them."""
    planes = T.to_planes(T.synthetic_image(300, 264))
    want = T.oracle_hot_path(planes, 2.0)
    enc.upload(planes)
    before = enc.stats()["encodes_with_redone_tiles"]
    dp = enc.enqueue(2.0, 0x1000)
    fr = enc.fetch_raw()
    st = enc.stats()
    assert st["tiles_redone_exact_roots"] == st["tiles"] == 5 * 5
    assert st["encodes_with_redone_tiles"] == before + 1
    frame = enc.assemble(fr, dp, 0)
    assert built.file_header(300, 264) + frame == T.assemble_codestream(want, 2.0)
    # an ordinary encode afterwards redoes nothing
    enc.enqueue(2.0, 0)
    enc.synchronize()
    st = enc.stats()
    assert st["tiles_redone_exact_roots"] == 0 and st["encodes_with_redone_tiles"] == before + 1


def test_one_hot_tile_is_redone_alone(built, enc):
    """A real overflow of the root table: a 4096x4096 frame with one 8x8 block of samples far above 1.0 at a
    tiny distance.  Only the tiles that see it are redone (enc_ac_strategy.cc:118-126 takes a Sqrt per
    coefficient: the results must not depend on the table), the codestream equals the oracle's on crops around the
    block and elsewhere, and tile_kernel's time stays that of a frame without the block plus the time ONE tile
    takes on an otherwise idle chip (the redo launch: about 45 us) -- not twice the frame's, as when the whole
    pipeline was redone."""
    size, d = 4096, 0.1
    img = T.synthetic_image(size, size, seed=77)
    y0, x0 = 17 * 64 + 24, 23 * 64 + 16
    cold = img.copy()
    img[y0:y0 + 8, x0:x0 + 8:2] = 40.0  # checkerboard columns: coefficients of a few thousand quantisation steps
    planes, cold_planes = T.to_planes(img), T.to_planes(cold)

    def tile_ms(p):
        enc.upload(p)
        ts = []
        for _ in range(6):
            enc.enqueue(d, 0)
            enc.synchronize()
            ts.append(enc.kernel_times()["tile_kernel"])
        return min(ts[2:]), enc.stats()

    t_cold, st_cold = tile_ms(cold_planes)
    t_hot, st_hot = tile_ms(planes)
    assert st_cold["tiles_redone_exact_roots"] == 0
    assert 1 <= st_hot["tiles_redone_exact_roots"] <= 4, st_hot
    assert t_hot <= 1.05 * t_cold + 0.06, (t_hot, t_cold)
    # bytes: the group that holds the block, a neighbour and far ones, against the oracle on the same crops
    got = built.HotPathOutput(enc.fetch_raw())
    _check_sampled_groups(planes, got, d, False, [(y0 // 256, x0 // 256), (y0 // 256, x0 // 256 + 1), (0, 0), (15, 15)])


def _random_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        w = int(rng.integers(9, 900))
        h = int(rng.integers(9, 700))
        d = float(np.round(10 ** rng.uniform(-1.3, 1.2), 3))  # 0.05 .. 16
        cases.append((w, h, d, int(rng.integers(0, 1 << 30))))
    return cases


@pytest.mark.parametrize("w,h,distance,seed", _random_cases(10, 20260902))
def test_random_frames_codestream_equals_oracle(built, w, h, distance, seed):
    """Random geometries (partial tiles, stripes and groups on both axes), distances across the
    supported range and per-case pixel seeds: the drop-in's codestream equals the oracle's."""
    img = T.synthetic_image(w, h, seed=seed, hard=(seed % 3 == 0))
    planes = T.to_planes(img)
    want = T.assemble_codestream(T.oracle_hot_path(planes, distance), distance)
    assert built.encode_file(planes, distance) == want


def _judge_cases():
    import test_oracle_known_answers as K
    cases = [("r2",) + k for k in sorted(K.JUDGE_R2)] + [("r3",) + k for k in sorted(K.JUDGE_R3)]
    cases += [("r4", k[1], k[2], k[3], k[4], k[0]) for k in sorted(K.JUDGE_R4)]
    cases += [("r5", k[1], k[2], k[3], k[4], k[0]) for k in sorted(K.JUDGE_R5)]
    return cases


@pytest.mark.parametrize("case", _judge_cases(), ids=lambda c: "%s_%dx%d_d%g_s%d_%s" % c)
def test_product_bytes_equal_the_judges_stand_in_builds(built, case):
    """The drop-in's codestream, reference-bytes mode, against the size + sha-256 of the bytes the unmodified
    reference sources produced in the judges' stand-in builds (VERDICT.md rounds 2, 3, 4 and 5) -- compared with the
    recorded hashes directly, no oracle in between.  (Round 5's distances go through EncodeFile's own clamp, which
    encode_file applies like the reference's enc_file.cc:63-65.)"""
    import hashlib
    import test_oracle_known_answers as K
    which, w, h, d, seed, kind = case
    if which == "r2":
        img = T.synthetic_image(w, h, seed=seed, hard=kind)
        want = K.JUDGE_R2[(w, h, d, seed, kind)]
    elif which == "r3":
        img = K.judge_r3_image(w, h, seed, kind)
        want = K.JUDGE_R3[(w, h, d, seed, kind)]
    elif which == "r4":
        img = K.judge_r4_image(kind, w, h, seed)
        want = K.JUDGE_R4[(kind, w, h, d, seed)]
    else:
        img = K.judge_r5_image(kind, w, h, seed)
        want = K.JUDGE_R5[(kind, w, h, d, seed)]
    built.emulate_reference_single_symbol_codes(True)
    try:
        got = built.encode_file(T.to_planes(img), d)
    finally:
        built.emulate_reference_single_symbol_codes(False)
    assert (len(got), hashlib.sha256(got).hexdigest()) == want


def test_static_constant_emulation_on_gpu(built, enc):
    """jxlt_set_strategy_distance / jxl::EmulateReferenceStaticConstants: a frame at distance 0.5
    of a process whose first frame was encoded at distance 16 (reference quirk,
    enc_ac_strategy.cc:178-185)."""
    planes = T.to_planes(T.synthetic_image(200, 137))
    plain = T.oracle_hot_path(planes, 0.5)
    T.set_strategy_distance(16.0)
    try:
        want = T.oracle_hot_path(planes, 0.5)
    finally:
        T.set_strategy_distance(0.0)
    assert (want.strategy != plain.strategy).any()  # the quirk matters for this frame
    jxl_want = T.assemble_codestream(want, 0.5)
    # C ABI
    enc.set_strategy_distance(16.0)
    try:
        got = enc.hot_path(planes, 0.5)
    finally:
        enc.set_strategy_distance(0.0)
    assert T.compare_results(want, got, "oracle", "gpu", check_debug=False) == []
    # host switch: the first frame of the "process" latches 16
    built.emulate_reference_static_constants(True)
    try:
        built.encode_file(T.to_planes(T.synthetic_image(64, 64)), 16.0)
        assert built.encode_file(planes, 0.5) == jxl_want
    finally:
        built.emulate_reference_static_constants(False)
    assert built.encode_file(planes, 0.5) == T.assemble_codestream(plain, 0.5)


def test_frame_batch_on_two_contexts(built):
    """BASELINE config #5 in miniature: a batch of 3840x2160 frames (135 groups, bottom group row
    112 px), encoded through two device contexts driven by two host threads so that one frame's
    upload overlaps the other's encode; every codestream equals the oracle's."""
    import threading
    frames = [T.to_planes(T.synthetic_image(3840, 2160, seed=100 + i)) for i in range(4)]
    want = [T.assemble_codestream(T.oracle_hot_path(p, 1.0), 1.0) for p in frames]
    got = [None] * len(frames)
    errors = []

    def worker(k):
        try:
            enc = built.Encoder(0)
            for i in range(k, len(frames), 2):
                enc.upload(frames[i])
                got[i] = enc.encode_resident(1.0, copy=True)
            enc.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(len(frames)):
        assert got[i] == want[i], "frame %d" % i


@pytest.mark.gpu
def test_frame_batch_encoder(built):
    """jxlt_batch_encoder_*: frames of different geometry, planar and raw-PFM-payload sources, pinned and
    ordinary memory, more frames than lanes; every codestream equals the oracle's."""
    sizes = [(3840, 2160), (200, 137), (1030, 520), (256, 256), (3840, 2160), (700, 300), (64, 72)]
    frames, want = [], []
    keep = []
    for i, (w, h) in enumerate(sizes):
        planes = T.to_planes(T.synthetic_image(w, h, seed=300 + i))
        want.append(T.assemble_codestream(T.oracle_hot_path(planes, 2.0), 2.0))
        if i % 3 == 1:  # PFM payload: interleaved, bottom row first, big endian
            payload = np.ascontiguousarray(planes.transpose(1, 2, 0)[::-1]).astype(">f4")
            frames.append((payload.view(np.uint8).reshape(-1), w, h, True))
        elif i % 3 == 2:  # page-locked planes with a row pitch
            arr, owner = built.pinned_empty((3, h, w + 24))
            arr[:, :, :w] = planes
            keep.append(owner)
            frames.append(arr[:, :, :w])
        else:
            frames.append(planes)
    import torch
    dev_frame = torch.from_numpy(T.to_planes(T.synthetic_image(520, 300, seed=77))).to("cuda:0")
    frames.append(dev_frame)  # a frame that already is in device memory (read in place)
    want.append(T.assemble_codestream(T.oracle_hot_path(dev_frame.cpu().numpy(), 2.0), 2.0))
    sizes = sizes + [(520, 300)]
    enc = built.BatchEncoder(0, lanes=3)
    got = enc.encode(frames, 2.0)
    again = enc.encode(frames[:2], 2.0)  # the encoder is reusable
    enc.close()
    for i in range(len(sizes)):
        assert got[i] == want[i], "frame %d %s" % (i, sizes[i])
    assert again == want[:2]
    # the multi-device form (BASELINE config #5's shape: one queue, lanes on every listed GPU; here GPU 0 twice)
    enc = built.BatchEncoder(lanes=2, devices=[0, 0])
    got = enc.encode(frames, 2.0)
    enc.close()
    for i in range(len(sizes)):
        assert got[i] == want[i], "multi-device frame %d %s" % (i, sizes[i])
    # error behaviour: a frame without a source is rejected, the others are still encoded
    enc = built.BatchEncoder(0, lanes=2)
    descs, k = enc.describe(frames[:2])
    descs[1].pfm_payload = None
    with pytest.raises(built.JxlTinyError):
        enc.run_described(descs, 2, 2.0)
    enc.close()
