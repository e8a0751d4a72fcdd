"""Every output must be decodable (BASELINE metric: "output djxl-decodable").  There is no djxl
in this image, so the codestreams are read back by tests/jxl_decoder.py, a reader written from
the codestream format (headers, TOC, prefix codes, context maps, the transmitted context tree,
VarDCT tokens, dequantisation, inverse transforms, inverse XYB).  It checks the syntax to the
last bit of every TOC section and the *meaning* of the stream: the reconstruction must be close
to the encoder's input, closer the smaller the distance -- a property that holds for any image
size and that neither the oracle nor the kernels know anything about."""
import numpy as np
import pytest

import jxl_decoder as D
import jxlt_testlib as T


def _encode_on_cpu(planes, distance, force_dct8=False):
    r = T.oracle_hot_path(planes, distance, force_dct8)
    return r, T.assemble_codestream(r, distance)


def _check_side_info(dec, r):
    assert np.array_equal(dec.quant_dc_image, r.quant_dc)
    assert np.array_equal(dec.raw_quant, r.raw_quant)
    first = (r.strategy & 1) == 1
    code = np.array([0, 6, 7])[r.strategy >> 1]
    assert np.array_equal(dec.strategy[first], code[first])
    assert np.array_equal(dec.cfl[0], r.ytox) and np.array_equal(dec.cfl[1], r.ytob)
    assert dec.num_tokens == sum(len(g) for g in r.group_tokens) // 3


CASES = [
    # w, h, distance, hard, minimum PSNR (dB, cube-root LMS domain)
    (9, 7, 1.0, False, 30.0),        # one group, partial blocks
    (64, 64, 1.0, False, 36.0),
    (200, 137, 0.1, False, 45.0),
    (200, 137, 1.0, False, 36.5),
    (200, 137, 16.0, False, 34.0),
    (264, 260, 8.0, False, 34.5),    # four groups; the reference's own bytes for this one are undecodable
    (520, 300, 2.0, False, 35.5),
    (72, 40, 0.1, True, 36.0),       # uniform noise: token-heavy
    (2100, 300, 1.0, False, 36.5),   # two DC groups
]


@pytest.mark.parametrize("w,h,distance,hard,min_psnr", CASES)
def test_codestream_decodes_and_resembles_the_input(built, w, h, distance, hard, min_psnr):
    planes = T.to_planes(T.synthetic_image(w, h, hard=hard))
    r, jxl = _encode_on_cpu(planes, distance)
    dec = D.decode(jxl)
    assert (dec.xsize, dec.ysize) == (w, h)
    assert dec.tree_leaves == 45
    _check_side_info(dec, r)
    assert D.psnr_opsin_db(planes, dec.linear_rgb) >= min_psnr


def test_frame_whose_corner_dc_group_is_one_block_decodes(built):
    """2049 x 2049: four DC groups, the last of them a single 8 x 8 block (its colour-correlation map, its DC image
    and its AC metadata are 1 x 1).  The reference traps on this shape (enc_frame.cc:335-339); the oracle
    -- and with it the product, tests/test_gpu_parity.py -- writes a stream that reads back to the last bit."""
    planes = T.to_planes(T.synthetic_image(2049, 2049, seed=102))
    jxl = bytes(T.oracle_encode_file(planes, 0.7, nthreads=8)[0])
    dec = D.decode(jxl)
    assert (dec.xsize, dec.ysize) == (2049, 2049)
    assert D.psnr_opsin_db(planes, dec.linear_rgb) > 37.0


def test_quality_follows_distance(built):
    planes = T.to_planes(T.synthetic_image(264, 200))
    psnr, size = [], []
    for d in (0.1, 0.5, 2.0, 8.0):
        _, jxl = _encode_on_cpu(planes, d)
        psnr.append(D.psnr_opsin_db(planes, D.decode(jxl).linear_rgb))
        size.append(len(jxl))
    assert psnr == sorted(psnr, reverse=True) and size == sorted(size, reverse=True), (psnr, size)
    assert psnr[0] > 45.0


@pytest.mark.parametrize("kind", ["rows", "columns", "diagonal", "noise"])
@pytest.mark.parametrize("force_dct8", [True, False])
def test_transform_orientation(built, kind, force_dct8):
    """Directional content at a small distance: a transposed coefficient layout of any of the three
    transforms (or of their scan orders / dequantisation tables) would wreck these."""
    h = w = 128
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    f = {"rows": 0.5 + 0.3 * np.sin(y * 1.3), "columns": 0.5 + 0.3 * np.sin(x * 1.3),
         "diagonal": 0.5 + 0.3 * np.sin(x * 0.9 + y * 0.4),
         "noise": 0.5 + 0.1 * np.random.default_rng(5).standard_normal((h, w))}[kind]
    planes = np.stack([f, 0.8 * f, 0.6 * f]).astype(np.float32)
    r, jxl = _encode_on_cpu(planes, 0.1, force_dct8)
    dec = D.decode(jxl)
    _check_side_info(dec, r)
    if not force_dct8 and kind in ("rows", "columns"):
        assert (dec.strategy == (7 if kind == "rows" else 6)).sum() > 100  # the search picks the long transform
    assert D.psnr_opsin_db(planes, dec.linear_rgb) >= 48.0


def test_reference_single_symbol_quirk(built):
    """With jxl::EmulateReferenceSingleSymbolCodes the bytes are the reference's -- and unreadable when
    a clustered histogram has a single symbol; the default output of the same frame decodes."""
    planes = T.to_planes(T.synthetic_image(264, 260))
    r = T.oracle_hot_path(planes, 8.0)
    good = T.assemble_codestream(r, 8.0)
    with T.reference_single_symbol_codes():
        ref_bytes = T.assemble_codestream(r, 8.0)
    assert ref_bytes != good and len(ref_bytes) >= len(good)
    D.decode(good)
    with pytest.raises(D.DecodeError):
        D.decode(ref_bytes)
    # where no code is degenerate the two modes give the same bytes
    planes = T.to_planes(T.synthetic_image(200, 137))
    r = T.oracle_hot_path(planes, 1.0)
    with T.reference_single_symbol_codes():
        ref_bytes = T.assemble_codestream(r, 1.0)
    assert ref_bytes == T.assemble_codestream(r, 1.0)


def test_damaged_streams_are_rejected(built):
    planes = T.to_planes(T.synthetic_image(300, 264))
    _, jxl = _encode_on_cpu(planes, 1.0)
    D.decode(jxl)
    for bad in (jxl[:-7], jxl + b"\0" * 3, jxl[:40] + jxl[41:], b"\xff\x0b" + jxl[2:]):
        with pytest.raises(D.DecodeError):
            D.decode(bad)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,distance", [(2100, 300, 1.0), (700, 520, 2.0), (256, 256, 0.5)])
def test_gpu_output_decodes(built, w, h, distance):
    planes = T.to_planes(T.synthetic_image(w, h, seed=77))
    jxl = built.encode_file(planes, distance)
    dec = D.decode(jxl)
    assert (dec.xsize, dec.ysize) == (w, h)
    assert D.psnr_opsin_db(planes, dec.linear_rgb) >= 36.0


@pytest.mark.gpu
def test_gpu_pfm_file_output_decodes(built, tmp_path):
    img = T.synthetic_image(520, 300, seed=5)
    path = tmp_path / "in.pfm"
    T.write_pfm(path, img, big_endian=True)
    jxl = built.encode_pfm_file(path, 1.0)
    dec = D.decode(jxl)
    assert D.psnr_opsin_db(T.to_planes(img), dec.linear_rgb) >= 36.5
