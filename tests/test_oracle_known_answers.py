"""Pins the oracle (and the host bitstream back-end) against the only reference
outputs that exist: the codestream *sizes* a one-off probe build of the
unmodified reference produced in the survey container.

The reference cannot be built in this image, so these are
size-only known answers, not golden bytes: the oracle stays "parity unpinned".
Observed: the unfused variant of the arithmetic model reproduces every probe
size; the canonical fused model (what a real AVX2 Highway build computes)
reproduces all but 1024x1024, where it is one byte longer -- bisected to the
fused multiply-adds of the XYB stage ("Oracle pinning").
"""
import ctypes as C

import pytest

import jxlt_testlib as T

# (w, h) -> bytes at distance 1.0 for the >= 4-lane builds of the probe.
PROBE_SIZES = {(9, 7): 190, (200, 137): 3463, (256, 256): 6702, (2100, 300): 60260, (1024, 1024): 97860}


def _size(w, h, lib=None):
    planes = T.to_planes(T.synthetic_image(w, h))
    saved = T._oracle
    try:
        if lib is not None:
            T._oracle = lib
        res = T.oracle_hot_path(planes, 1.0)
    finally:
        T._oracle = saved
    with T.reference_single_symbol_codes():  # the probe sizes are outputs of the reference itself
        return len(T.assemble_codestream(res, 1.0))


@pytest.mark.parametrize("wh", [(9, 7), (200, 137), (256, 256), (2100, 300)])
def test_canonical_model_reproduces_probe_sizes(built, wh):
    assert _size(*wh) == PROBE_SIZES[wh]


def test_canonical_model_1024_is_within_one_byte(built):
    assert abs(_size(1024, 1024) - PROBE_SIZES[(1024, 1024)]) <= 1


@pytest.mark.parametrize("wh", sorted(PROBE_SIZES))
def test_unfused_variant_reproduces_all_probe_sizes(built, wh):
    base = T.oracle()
    lib = C.CDLL(str(T.ROOT / "oracle" / "liboracle_nofma.so"))
    lib.orc_encode_hot_path.argtypes = base.orc_encode_hot_path.argtypes
    lib.orc_encode_hot_path.restype = C.c_int
    lib.orc_frame_free.argtypes = base.orc_frame_free.argtypes
    lib.orc_compute_distance_params.argtypes = base.orc_compute_distance_params.argtypes
    assert _size(*wh, lib=lib) == PROBE_SIZES[wh]


def test_codestream_starts_with_signature(built):
    planes = T.to_planes(T.synthetic_image(64, 64))
    cs = T.assemble_codestream(T.oracle_hot_path(planes, 1.0), 1.0)
    assert cs[:2] == b"\xff\x0a"


def test_single_block_images_are_rejected(built):
    # The reference traps on images that fit one 8x8 block.
    import numpy as np
    planes = np.zeros((3, 8, 8), np.float32)
    with pytest.raises(ValueError):
        T.oracle_hot_path(planes, 1.0)
