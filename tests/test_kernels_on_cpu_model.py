"""Runs the product's HIP kernels (libjxl-tiny_amd/csrc/jxlt_device.h) on the
fiber-based CPU execution model in tests/hipsim and requires bit-exact equality
with the oracle for every intermediate and output.  This is test infrastructure
only -- the product never runs on the CPU -- and lets kernel changes be checked
in a container without a GPU; the `-m gpu` tests repeat the comparison on a real
MI355X through the C ABI."""
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
]


@pytest.mark.parametrize("w,h,distance,hard,dct8", CASES)
def test_kernels_match_oracle_bit_exact(built, w, h, distance, hard, dct8):
    planes = T.to_planes(T.synthetic_image(w, h, hard=hard))
    want = T.oracle_hot_path(planes, distance, dct8)
    got = T.sim_hot_path(planes, distance, dct8)
    assert T.compare_results(want, got, "oracle", "kernels") == []
